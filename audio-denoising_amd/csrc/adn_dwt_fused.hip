// Fused tile kernels for the mono VisuShrink block denoiser (placeholder: not handled yet).
#include "adn_kernels.hpp"

namespace adn {

template <typename T>
int wavelet_denoise_fused(Ctx *c, const T *d_in, T *d_out, int64_t n_blocks,
                          const int64_t *d_in_off, const int64_t *d_frames,
                          const int64_t *d_out_off, int64_t max_frames, int wavelet, int level,
                          int thr_mode, double *d_sigma, double *d_thresh, int64_t *d_zeroed,
                          bool *handled) {
    *handled = false;
    return ADN_OK;
}

template int wavelet_denoise_fused<float>(Ctx *, const float *, float *, int64_t, const int64_t *, const int64_t *, const int64_t *, int64_t, int, int, int, double *, double *, int64_t *, bool *);
template int wavelet_denoise_fused<double>(Ctx *, const double *, double *, int64_t, const int64_t *, const int64_t *, const int64_t *, int64_t, int, int, int, double *, double *, int64_t *, bool *);

}  // namespace adn
