#include "adn_kernels.hpp"
namespace adn {
template <typename T>
int noise_psd(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, int, int, int, double, double, int, int, T *) { set_error("not built yet"); return ADN_ERR_UNSUPPORTED; }
template <typename T>
int spectral_subtract(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, const T *, int64_t, int, int, double, double, int, int, T *, const int64_t *) { set_error("not built yet"); return ADN_ERR_UNSUPPORTED; }
template <typename T>
int wiener(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, int, int, int, double, int, int, T *, const int64_t *, T *) { set_error("not built yet"); return ADN_ERR_UNSUPPORTED; }
#define I(T) \
template int noise_psd<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, int, int, int, double, double, int, int, T *); \
template int spectral_subtract<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, const T *, int64_t, int, int, double, double, int, int, T *, const int64_t *); \
template int wiener<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, int, int, int, double, int, int, T *, const int64_t *, T *);
I(float) I(double)
}
