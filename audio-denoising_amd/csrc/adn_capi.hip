// extern "C" surface of libadn_hip.so: context management, length helpers and
// thin typed wrappers around the templated implementations.
#include <math.h>
#include <string.h>

#include "adn_kernels.hpp"

namespace adn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int aux_reserve(Ctx *c, size_t bytes) {
    if (c->ws_fixed) {      // also when an earlier ordinary call left a large enough buffer: same answer every time
        set_error("this mode needs %zu bytes of device memory beyond the fixed scratch buffer; "
                  "the entry point that allocates nothing does not support it", bytes);
        return ADN_ERR_NOMEM;
    }
    if (bytes <= c->aux_bytes) return ADN_OK;
    if (c->aux) {
        ADN_HIP(hipStreamSynchronize(c->stream));
        ADN_HIP(hipFree(c->aux));
        c->aux = nullptr;
        c->aux_bytes = 0;
    }
    void *p = nullptr;
    const size_t want = bytes + bytes / 8 + (1u << 20);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        set_error("allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return ADN_ERR_NOMEM;
    }
    c->aux = static_cast<char *>(p);
    c->aux_bytes = want;
    return ADN_OK;
}

int ws_reserve(Ctx *c, size_t bytes) {
    if (bytes <= c->ws_bytes) return ADN_OK;
    if (c->ws_fixed) {
        set_error("input too large for the entry point that allocates nothing: it needs %zu bytes of "
                  "scratch, the fixed scratch buffer holds %zu", bytes, c->ws_bytes);
        return ADN_ERR_NOMEM;
    }
    // round up generously so that slowly growing batches do not reallocate every call
    size_t want = bytes + bytes / 8 + (1u << 20);
    if (c->ws) {
        ADN_HIP(hipStreamSynchronize(c->stream));
        ADN_HIP(hipFree(c->ws));
        c->ws = nullptr;
        c->ws_bytes = 0;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return ADN_ERR_NOMEM;
    }
    c->ws = static_cast<char *>(p);
    c->ws_bytes = want;
    return ADN_OK;
}

void drop_kernel_records(Ctx *c) {
    for (KernelRec &r : c->krecs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    c->krecs.clear();
}

}  // namespace adn

template <typename T>
static int mad_impl(adn_ctx *c, const T *d_x, int64_t n_seg, const int64_t *d_off, const int64_t *d_len,
                    int64_t max_len, double *d_median, double *d_mad) {
    ADN_REQUIRE(c && d_x && d_off && d_len && d_mad, "null pointer argument");
    ADN_REQUIRE(n_seg >= 0 && max_len >= 1, "n_seg must be >= 0 and max_len >= 1");
    if (n_seg == 0) return ADN_OK;
    adn::DeviceGuard dg(c->device);
    adn::ws_reset(c);
    ADN_TRY(adn::ws_reserve(c, adn::ws_need(n_seg, 8) + 2 * adn::select_ws_bytes(n_seg, max_len)));
    double *med = d_median ? d_median : adn::ws_take<double>(c, n_seg);
    adn::SelectArgs a;
    a.n_seg = n_seg;
    a.d_seg_off = d_off;
    a.d_seg_len = d_len;
    a.max_len = max_len;
    a.mode = adn::SEL_MEDIAN;
    a.d_out = med;
    ADN_TRY(adn::launch_select<T>(c, d_x, a));
    a.xform = adn::XF_ABSDEV;
    a.d_center = med;
    a.d_out = d_mad;
    return adn::launch_select<T>(c, d_x, a);
}

using adn::Ctx;

// Scratch of the entry point that allocates nothing: one copy per GPU, alive until the process
// ends.  In host code this name is only a handle for hipGetSymbolAddress; the device address of
// each GPU's copy is looked up once in adn_create and kept in the context (Ctx::scratch).
static __device__ __align__(256) unsigned char scratch[ADN_NOALLOC_SCRATCH_BYTES];

extern "C" {

const char *adn_last_error(void) { return adn::g_err; }

int adn_abi_version(void) { return ADN_ABI_VERSION; }

int adn_version(void) { return 100; }

int adn_create(adn_ctx **out, int device) {
    ADN_REQUIRE(out != nullptr, "ctx output pointer is null");
    int count = 0;
    ADN_HIP(hipGetDeviceCount(&count));
    ADN_REQUIRE(device >= 0 && device < count, "device %d out of range (%d visible)", device, count);
    adn::DeviceGuard dg(device);
    hipDeviceProp_t prop;
    ADN_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        adn::set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return ADN_ERR_UNSUPPORTED;
    }
    void *scratch_dev = nullptr;          // this device is current (DeviceGuard above)
    ADN_HIP(hipGetSymbolAddress(&scratch_dev, HIP_SYMBOL(scratch)));
    adn_ctx *c = new adn_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    c->scratch = static_cast<char *>(scratch_dev);
    *out = c;
    return ADN_OK;
}

int adn_destroy(adn_ctx *c) {
    if (!c) return ADN_OK;
    adn::DeviceGuard dg(c->device);
    (void)hipStreamSynchronize(c->stream);
    adn::drop_kernel_records(c);
    if (c->ws) (void)hipFree(c->ws);
    if (c->aux) (void)hipFree(c->aux);
    for (int i = 0; i < c->n_tables; ++i)
        if (c->tables[i].dev) (void)hipFree(c->tables[i].dev);
    delete c;
    return ADN_OK;
}

int adn_set_stream(adn_ctx *c, void *s) {
    ADN_REQUIRE(c != nullptr, "ctx is null");
    c->stream = static_cast<hipStream_t>(s);
    return ADN_OK;
}

int adn_set_option(adn_ctx *c, int option, int value) {
    ADN_REQUIRE(c != nullptr, "ctx is null");
    ADN_REQUIRE(option >= ADN_OPT_FORCE_GENERIC && option <= ADN_OPT_KERNEL_TIMING, "unknown option %d", option);
    if (option == ADN_OPT_FORCE_GENERIC) c->force_generic = value != 0;
    if (option == ADN_OPT_FORCE_FALLBACK) c->force_fallback = value != 0;
    if (option == ADN_OPT_ABLATE) c->ablate = value;
    if (option == ADN_OPT_KERNEL_TIMING) {
        adn::DeviceGuard dg(c->device);
        if (!value) adn::drop_kernel_records(c);
        c->ktime = value != 0;
    }
    return ADN_OK;
}

int adn_kernel_timing_only(adn_ctx *c, const char *name) {
    ADN_REQUIRE(c != nullptr, "ctx is null");
    c->ktime_only = name ? name : "";
    return ADN_OK;
}

int64_t adn_kernel_times(adn_ctx *c, char *buf, int64_t cap) {
    if (!c || !buf || cap < 1) return -1;
    adn::DeviceGuard dg(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    // aggregate by launch-site name, in order of first appearance
    std::vector<const char *> names;
    std::vector<double> total;
    std::vector<long long> count;
    for (const adn::KernelRec &r : c->krecs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
        size_t i = 0;
        for (; i < names.size(); ++i)
            if (strcmp(names[i], r.name) == 0) break;
        if (i == names.size()) {
            names.push_back(r.name);
            total.push_back(0.0);
            count.push_back(0);
        }
        total[i] += ms;
        count[i] += 1;
    }
    std::string out;
    char line[384];
    for (size_t i = 0; i < names.size(); ++i) {
        snprintf(line, sizeof(line), "%s\t%lld\t%.6f\n", names[i], count[i], total[i]);
        out += line;
    }
    if ((int64_t)out.size() + 1 > cap) return -(int64_t)(out.size() + 1);     // the records stay: call again with room
    adn::drop_kernel_records(c);
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int64_t)out.size();
}

int adn_synchronize(adn_ctx *c) {
    ADN_REQUIRE(c != nullptr, "ctx is null");
    adn::DeviceGuard dg(c->device);
    ADN_HIP(hipStreamSynchronize(c->stream));
    return ADN_OK;
}

int64_t adn_workspace_bytes(adn_ctx *c) { return c ? (int64_t)c->ws_bytes : 0; }

int adn_select_resolve(adn_ctx *c, const uint32_t *d_hist, int64_t n_seg, int key_bits, int done_bits, int64_t *d_rank,
                       uint64_t *d_prefix) {
    return adn::select_resolve(c, d_hist, n_seg, key_bits, done_bits, reinterpret_cast<long long *>(d_rank),
                               reinterpret_cast<unsigned long long *>(d_prefix));
}

int64_t adn_wavelet_out_frames(int64_t frames) { return 2 * ((frames + 1) / 2); }

int64_t adn_denoise_block_frames(int64_t rate, int64_t frames) {
    if (rate <= 0) return 0;
    const double duration = (double)frames / (double)rate;
    return (int64_t)((double)rate * duration * 0.10);
}

int64_t adn_stft_num_frames(int64_t n, int n_fft, int hop, int padded) {
    if (n < n_fft || n_fft <= 0 || hop <= 0) return -1;
    int64_t nadd = 0;
    if (padded) {
        int64_t r = (n - n_fft) % hop;
        nadd = ((hop - r) % hop) % n_fft;
    }
    return 1 + (n + nadd - n_fft) / hop;
}

int64_t adn_istft_out_samples(int64_t n, int n_fft, int hop) {
    const int64_t T = adn_stft_num_frames(n, n_fft, hop, 1);
    return T < 0 ? -1 : (T - 1) * hop;
}

#define ADN_WRAP_T(SUF, T)                                                                          \
    int adn_wavelet_denoise_##SUF(adn_ctx *c, const T *d_in, T *d_out, int64_t n_blocks,            \
                                  const int64_t *d_in_off, const int64_t *d_frames,                 \
                                  const int64_t *d_out_off, int64_t max_frames, int n_ch,           \
                                  int axis_mode, int wavelet, int level, int ext_mode,              \
                                  int thr_mode, double *d_sigma, double *d_thresh,                  \
                                  int64_t *d_zeroed) {                                              \
        return adn::wavelet_denoise<T>(c, d_in, d_out, n_blocks, d_in_off, d_frames, d_out_off,     \
                                       max_frames, n_ch, axis_mode, wavelet, level, ext_mode,       \
                                       thr_mode, d_sigma, d_thresh, d_zeroed);                      \
    }                                                                                               \
    int adn_mad_##SUF(adn_ctx *c, const T *d_x, int64_t n_seg, const int64_t *d_off,                \
                      const int64_t *d_len, int64_t max_len, double *d_median, double *d_mad) {     \
        return mad_impl<T>(c, d_x, n_seg, d_off, d_len, max_len, d_median, d_mad);                  \
    }                                                                                               \
    int adn_wavedec_##SUF(adn_ctx *c, const T *d_x, int64_t n, int wavelet, int level, int ext,     \
                          T *d_coeffs, int64_t cap, int64_t *h_band_len) {                          \
        return adn::wavedec<T>(c, d_x, n, wavelet, level, ext, d_coeffs, cap, h_band_len);          \
    }                                                                                               \
    int adn_waverec_##SUF(adn_ctx *c, const T *d_coeffs, const int64_t *h_band_len, int wavelet,    \
                          int level, int ext, T *d_y, int64_t cap, int64_t *h_n_out) {              \
        return adn::waverec<T>(c, d_coeffs, h_band_len, wavelet, level, ext, d_y, cap, h_n_out);    \
    }                                                                                               \
    int adn_noise_psd_##SUF(adn_ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,          \
                            const int64_t *d_len, int64_t max_len, int n_ch, int n_fft, int hop,    \
                            int padded, double percentile, double spectral_floor, int freq_bins,    \
                            int time_frames, T *d_baseline) {                                       \
        return adn::noise_psd<T>(c, d_x, n_sig, d_off, d_len, max_len, n_ch, n_fft, hop, padded,    \
                                 percentile, spectral_floor, freq_bins, time_frames, d_baseline);   \
    }                                                                                               \
    int adn_hpss_##SUF(adn_ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,               \
                       const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate, int n_fft,  \
                       int hop, int kernel_size, double margin_h, double margin_p, double power,     \
                       int mask_type, int n_bands, const double *h_lo_hz, const double *h_hi_hz,     \
                       int preserve_transients, double threshold, double blend, T *d_yh, T *d_yp,    \
                       const int64_t *d_y_off, double *d_stats) {                                    \
        return adn::hpss_separate<T>(c, d_x, n_sig, d_off, d_len, max_len, n_ch, sample_rate, n_fft, \
                                     hop, kernel_size, margin_h, margin_p, power, mask_type, n_bands, \
                                     h_lo_hz, h_hi_hz, preserve_transients, threshold, blend, d_yh,  \
                                     d_yp, d_y_off, d_stats);                                        \
    }                                                                                               \
    int adn_band_summary_##SUF(adn_ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,       \
                               const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate,    \
                               int n_fft, int hop, int padded, int n_bands, const double *h_lo_hz,  \
                               const double *h_hi_hz, int n_perc, const double *h_perc,             \
                               double *d_mean_mag, double *d_perc_db, T *d_time_mean, T *d_mag) {   \
        return adn::band_summary<T>(c, d_x, n_sig, d_off, d_len, max_len, n_ch, sample_rate, n_fft, \
                                    hop, padded, n_bands, h_lo_hz, h_hi_hz, n_perc, h_perc,         \
                                    d_mean_mag, d_perc_db, d_time_mean, d_mag);                     \
    }                                                                                               \
    int adn_spectral_subtract_##SUF(adn_ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,  \
                                    const int64_t *d_len, int64_t max_len, int n_ch,                \
                                    const T *d_noise, int64_t noise_stride, int n_fft, int hop,     \
                                    double alpha, double beta, int smooth_time, int smooth_freq,    \
                                    T *d_y, const int64_t *d_y_off, int16_t *d_pcm16) {             \
        return adn::spectral_subtract<T>(c, d_x, n_sig, d_off, d_len, max_len, n_ch, d_noise,       \
                                         noise_stride, n_fft, hop, alpha, beta, smooth_time,        \
                                         smooth_freq, d_y, d_y_off, d_pcm16);                       \
    }                                                                                               \
    int adn_wiener_##SUF(adn_ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,             \
                         const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate,          \
                         int n_fft, int hop, double update_ms, int kernel_bins, int vad_guided,     \
                         T *d_y, const int64_t *d_y_off, T *d_gain, int16_t *d_pcm16) {             \
        return adn::wiener<T>(c, d_x, n_sig, d_off, d_len, max_len, n_ch, sample_rate, n_fft, hop,  \
                              update_ms, kernel_bins, vad_guided, d_y, d_y_off, d_gain, d_pcm16);   \
    }                                                                                               \
    int adn_wavelet_shrink_##SUF(adn_ctx *c, const T *d_in, T *d_out, int64_t n_blocks,             \
                                 const int64_t *d_in_off, const int64_t *d_frames,                  \
                                 const int64_t *d_out_off, int64_t max_frames, int wavelet,         \
                                 int level, int ext_mode, int thr_mode, const double *d_thresh,     \
                                 int64_t *d_zeroed) {                                               \
        return adn::wavelet_shrink<T>(c, d_in, d_out, n_blocks, d_in_off, d_frames, d_out_off,      \
                                      max_frames, wavelet, level, ext_mode, thr_mode, d_thresh,     \
                                      d_zeroed);                                                    \
    }                                                                                               \
    int adn_select_hist_##SUF(adn_ctx *c, const T *d_vals, int64_t n_seg, const int64_t *d_off,     \
                              const int64_t *d_len, int64_t max_len, int absdev,                    \
                              const double *d_center, const uint64_t *d_prefix, int done_bits,      \
                              uint32_t *d_hist) {                                                   \
        return adn::select_hist<T>(c, d_vals, n_seg, d_off, d_len, max_len, absdev, d_center,       \
                                   reinterpret_cast<const unsigned long long *>(d_prefix),          \
                                   done_bits, d_hist);                                              \
    }                                                                                               \
    int adn_select_value_##SUF(adn_ctx *c, const uint64_t *d_prefix, int64_t n_seg, double *d_out) { \
        return adn::select_value<T>(c, reinterpret_cast<const unsigned long long *>(d_prefix),      \
                                    n_seg, d_out);                                                  \
    }                                                                                               \
    int adn_select_combine_##SUF(adn_ctx *c, const double *d_lo, const double *d_hi, int64_t n_seg, \
                                 int64_t n_values, double q, double *d_out) {                       \
        return adn::select_combine<T>(c, d_lo, d_hi, n_seg, n_values, q, d_out);                    \
    }                                                                                               \
    int adn_stft_rows_##SUF(adn_ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,          \
                            const int64_t *d_len, int64_t max_len, int n_ch, int n_fft, int hop,    \
                            int padded, int power, T *d_S, int64_t row_stride) {                    \
        return adn::stft_rows<T>(c, d_x, n_sig, d_off, d_len, max_len, n_ch, n_fft, hop, padded,    \
                                 power, d_S, row_stride);                                           \
    }                                                                                               \
    int adn_vad_frames_##SUF(adn_ctx *c, const T *d_x, int64_t n_buf, int n_ch,                     \
                             int64_t base_sample, int64_t n_total, int sample_rate,                 \
                             double update_ms, int hop, int64_t t0, int64_t n_frames,               \
                             uint8_t *d_vad) {                                                      \
        return adn::vad_frames<T>(c, d_x, n_buf, n_ch, base_sample, n_total, sample_rate,           \
                                  update_ms, hop, t0, n_frames, d_vad);                             \
    }                                                                                               \
    int adn_rows_masked_sum_##SUF(adn_ctx *c, const T *d_S, int64_t n_rows, int64_t row_len,        \
                                  int64_t row_stride, const uint8_t *d_mask, double *d_sum,         \
                                  int64_t *d_count) {                                               \
        return adn::rows_masked_sum<T>(c, d_S, n_rows, row_len, row_stride, d_mask, d_sum,          \
                                       reinterpret_cast<long long *>(d_count));                     \
    }                                                                                               \
    int adn_wiener_gain_##SUF(adn_ctx *c, const double *d_noise_psd, const double *d_speech_psd,    \
                              int64_t n_sig, int n_bins, int kernel_bins, T *d_gain) {              \
        return adn::wiener_gain<T>(c, d_noise_psd, d_speech_psd, n_sig, n_bins, kernel_bins,        \
                                   d_gain);                                                         \
    }                                                                                               \
    int adn_stft_apply_gain_##SUF(adn_ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,    \
                                  const int64_t *d_len, int64_t max_len, int n_ch, int n_fft,       \
                                  int hop, const T *d_gain, int64_t gain_stride, T *d_y,            \
                                  const int64_t *d_y_off, int16_t *d_pcm16) {                       \
        return adn::stft_apply_gain<T>(c, d_x, n_sig, d_off, d_len, max_len, n_ch, n_fft, hop,      \
                                       d_gain, gain_stride, d_y, d_y_off, d_pcm16);                 \
    }                                                                                               \
    int adn_window_rms_##SUF(adn_ctx *c, const T *d_x, int64_t n_rows, int n_ch, int64_t win,       \
                             double *d_rms) {                                                       \
        return adn::window_rms<T>(c, d_x, n_rows, n_ch, win, d_rms);                                \
    }                                                                                               \
    int adn_pcm_encode_##SUF(adn_ctx *c, const T *d_x, int64_t n, int bits, void *d_q, T *d_xq) {   \
        return adn::pcm_encode<T>(c, d_x, n, bits, d_q, d_xq);                                      \
    }                                                                                               \
    int adn_synth_noisy_speech_##SUF(adn_ctx *c, T *d_x, int64_t n, int64_t start, int sample_rate, \
                                     uint64_t seed, uint64_t stream, double sigma,                  \
                                     double speech_gain) {                                          \
        return adn::synth_noisy_speech<T>(c, d_x, n, start, sample_rate, seed, stream, sigma,       \
                                          speech_gain);                                             \
    }

ADN_WRAP_T(f32, float)
ADN_WRAP_T(f64, double)

int adn_wavelet_denoise_noalloc_f32(adn_ctx *c, const float *d_in, float *d_out, int64_t n_blocks,
                                    const int64_t *d_in_off, const int64_t *d_frames,
                                    const int64_t *d_out_off, int64_t max_frames, int n_ch, int axis_mode,
                                    int wavelet, int level, int ext_mode, int thr_mode, double *d_sigma,
                                    double *d_thresh, int64_t *d_zeroed) {
    ADN_REQUIRE(c != nullptr, "ctx is null");
    ADN_REQUIRE(c->scratch != nullptr, "the static scratch buffer was not resolved for device %d", c->device);
    // the same kernels as adn_wavelet_denoise_f32, their workspace carved from this GPU's static
    // scratch instead of the growable allocation; requests beyond it fail in ws_reserve / aux_reserve
    char *const ws = c->ws;
    const size_t ws_bytes = c->ws_bytes;
    const bool ws_fixed = c->ws_fixed;
    c->ws = c->scratch;
    c->ws_bytes = ADN_NOALLOC_SCRATCH_BYTES;
    c->ws_fixed = true;
    const int rc = adn::wavelet_denoise<float>(c, d_in, d_out, n_blocks, d_in_off, d_frames, d_out_off, max_frames,
                                               n_ch, axis_mode, wavelet, level, ext_mode, thr_mode, d_sigma,
                                               d_thresh, d_zeroed);
    c->ws = ws;
    c->ws_bytes = ws_bytes;
    c->ws_fixed = ws_fixed;
    return rc;
}

}  // extern "C"
