// Internal launch interfaces between the translation units of libadn_hip.so.
#pragma once
#include "adn_common.hpp"

namespace adn {

// First element of sample run `run` of `nrun` runs of `len` consecutive elements over n elements (n >= nrun * len):
// the run lies in its own stratum [run n / nrun, (run + 1) n / nrun) at a position that a hash of the run index
// picks.  Evenly spaced runs would sample a periodic signal at one phase whenever n / nrun is close to a multiple
// of the period, and brackets taken from such a sample miss.
__host__ __device__ inline long long sample_run_start(long long n, int nrun, int len, int run) {
    const long long a = (long long)(((unsigned long long)run * (unsigned long long)n) / (unsigned long long)nrun);
    const long long b = (long long)(((unsigned long long)(run + 1) * (unsigned long long)n) / (unsigned long long)nrun);
    const long long room = b - a - len;
    unsigned int h = (unsigned int)run * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return a + (room > 0 ? (long long)(((unsigned long long)h * (unsigned long long)(room + 1)) >> 32) : 0);
}

// ------------------------------------------------ segmented order statistics
enum { SEL_MEDIAN = 0, SEL_PERCENTILE = 1, SEL_PERCENTILE_DB20 = 2,
       // internal (brackets taken from a sample of n values): the order statistic at rank (n - 1) q / 100 minus / plus
       // 4.5 standard deviations of that rank in a sample of n (and two ranks), clamped to the sample
       SEL_BRACKET_LO = 3, SEL_BRACKET_HI = 4 };
enum { XF_NONE = 0, XF_ABSDEV = 1 };

struct SelectArgs {
    int64_t n_seg = 0;
    const int64_t *d_seg_off = nullptr;   // element offset per segment, or null: seg * seg_stride
    int64_t seg_stride = 0;
    const int64_t *d_seg_len = nullptr;   // length per (seg / len_div), or null: uniform_len
    int64_t len_div = 1;
    int64_t uniform_len = 0;
    int64_t max_len = 0;                  // upper bound of any segment length (sizes LDS)
    int mode = SEL_MEDIAN;
    double q = 50.0;                      // percentile in [0, 100] for SEL_PERCENTILE
    int xform = XF_NONE;
    const double *d_center = nullptr;     // per segment, for XF_ABSDEV
    double *d_out = nullptr;              // per segment result
    // optional, short segments only: masked mean of each segment in the same pass
    const unsigned char *d_mask = nullptr;   // mask[(seg / mask_div) * mask_stride + i]
    int64_t mask_div = 1, mask_stride = 0;
    double *d_masked_mean = nullptr;
    // binary32 rows held in groups of four values (the layout rows_wave writes): value i of segment seg is
    // d_vals[(seg / blk_rows) * blk_sig_stride + (i / 4) * 4 * quad_stride + (seg % blk_rows) * 4 + i % 4];
    // quad_stride == 1: plain contiguous segments (the fields above)
    int64_t quad_stride = 1, blk_rows = 1, blk_sig_stride = 0;
    // grouped rows long enough for the streaming route may take it: the caller has reserved
    // select_rows_stream_ws_bytes of workspace for this call
    bool rows_stream = false;
};
size_t select_rows_stream_ws_bytes(const Ctx *c, int64_t n_sig, int64_t rows_per_sig, int64_t max_len, size_t value_bytes);
// longest row of the grouped layout when the caller sets rows_stream (the streaming route has its own exact fallback,
// so it reaches beyond the register-resident select)
int64_t select_grouped_limit_stream(const Ctx *c, size_t value_bytes);
// longest row of the grouped layout launch_select takes (values of value_bytes bytes)
int64_t select_grouped_limit(size_t value_bytes);
// longest segment launch_select handles inside one workgroup (longer ones take the
// multi-workgroup path, which has no fused masked mean)
int64_t select_short_limit();

// Exact numpy-compatible median / linearly interpolated percentile of every
// segment (NaN if the segment is empty or holds a NaN).
template <typename T>
int launch_select(Ctx *c, const T *d_vals, const SelectArgs &a);
// Workspace (bytes) a launch_select call carves out of the context's reserved workspace;
// callers add it to their ws_reserve request, once per launch_select call they make.
size_t select_ws_bytes(int64_t n_seg, int64_t max_len);

// Median of segments of which only a gathered SUBSET is at hand (the keys inside a value bracket): list s holds
// d_len[s] keys at d_cand + s * cap; the ranks are those of the original segment (d_orig_len[s] keys) minus
// d_below[s], the number of its keys below the bracket.  Lists with d_valid[s] == 0 are skipped (d_out untouched).
// cap <= select_subset_max_keys(sizeof key); lists that fit LDS are staged there whole, longer ones are bracketed
// again until they do (the caller reserves select_subset_ws_bytes of workspace per call for that).
struct SubsetSelect {
    int64_t n_seg = 0, cap = 0;
    const int64_t *d_len = nullptr, *d_orig_len = nullptr;
    const unsigned long long *d_below = nullptr;
    const int *d_valid = nullptr;
    int xform = XF_NONE;
    const double *d_center = nullptr;     // per segment, for XF_ABSDEV
    double *d_out = nullptr;              // per segment result
    double *d_pair = nullptr;             // optional: the two middle order statistics, [2*s], [2*s+1]
};
template <typename T>
int launch_select_subset(Ctx *c, const T *d_cand, const SubsetSelect &a);
size_t select_subset_max_keys(size_t key_bytes);
size_t select_subset_lds_keys(size_t key_bytes);   // lists up to this length are staged in LDS
size_t select_subset_ws_bytes(int64_t n_seg, int64_t cap, size_t key_bytes);

// -------------------------------------------------------------- wavelets
template <typename T>
int wavelet_denoise(Ctx *c, const T *d_in, T *d_out, int64_t n_blocks, const int64_t *d_in_off,
                    const int64_t *d_frames, const int64_t *d_out_off, int64_t max_frames,
                    int n_ch, int axis_mode, int wavelet, int level, int ext_mode, int thr_mode,
                    double *d_sigma, double *d_thresh, int64_t *d_zeroed);
template <typename T>
int wavedec(Ctx *c, const T *d_x, int64_t n, int wavelet, int level, int ext_mode, T *d_coeffs,
            int64_t cap, int64_t *h_band_len);
template <typename T>
int waverec(Ctx *c, const T *d_coeffs, const int64_t *h_band_len, int wavelet, int level,
            int ext_mode, T *d_y, int64_t cap, int64_t *h_n_out);

// ------------------------------------------------------------ STFT stages
template <typename T>
int noise_psd(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len,
              int64_t max_len, int n_ch, int n_fft, int hop, int padded, double percentile,
              double spectral_floor, int freq_bins, int time_frames, T *d_baseline);
template <typename T>
int spectral_subtract(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,
                      const int64_t *d_len, int64_t max_len, int n_ch, const T *d_noise,
                      int64_t noise_stride, int n_fft, int hop, double alpha, double beta,
                      int smooth_time, int smooth_freq, T *d_y, const int64_t *d_y_off, int16_t *d_pcm16);
template <typename T>
int wiener(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len,
           int64_t max_len, int n_ch, int sample_rate, int n_fft, int hop, double update_ms,
           int kernel_bins, int vad_guided, T *d_y, const int64_t *d_y_off, T *d_gain, int16_t *d_pcm16);

template <typename T>
int band_summary(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len,
                 int64_t max_len, int n_ch, int sample_rate, int n_fft, int hop, int padded, int n_bands,
                 const double *h_lo_hz, const double *h_hi_hz, int n_perc, const double *h_perc,
                 double *d_mean_mag, double *d_perc_db, T *d_time_mean, T *d_mag);

template <typename T>
int hpss_separate(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len, int64_t max_len,
                  int n_ch, int sample_rate, int n_fft, int hop, int kernel_size, double margin_h, double margin_p,
                  double power, int mask_type, int n_bands, const double *h_lo_hz, const double *h_hi_hz,
                  int preserve_transients, double threshold, double blend, T *d_yh, T *d_yp,
                  const int64_t *d_y_off, double *d_stats);

// ------------------------------------------- one recording split over devices
template <typename T>
int wavelet_shrink(Ctx *c, const T *d_in, T *d_out, int64_t n_blocks, const int64_t *d_in_off,
                   const int64_t *d_frames, const int64_t *d_out_off, int64_t max_frames, int wavelet, int level,
                   int ext_mode, int thr_mode, const double *d_thresh, int64_t *d_zeroed);
template <typename T>
int select_hist(Ctx *c, const T *d_vals, int64_t n_seg, const int64_t *d_off, const int64_t *d_len, int64_t max_len,
                int absdev, const double *d_center, const unsigned long long *d_prefix, int done_bits,
                unsigned int *d_hist);
int select_resolve(Ctx *c, const unsigned int *d_hist, int64_t n_seg, int key_bits, int done_bits, long long *d_rank,
                   unsigned long long *d_prefix);
template <typename T>
int select_value(Ctx *c, const unsigned long long *d_prefix, int64_t n_seg, double *d_out);
template <typename T>
int select_combine(Ctx *c, const double *d_lo, const double *d_hi, int64_t n_seg, int64_t n_values, double q,
                   double *d_out);
template <typename T>
int stft_rows(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len, int64_t max_len, int n_ch,
              int n_fft, int hop, int padded, int power, T *d_S, int64_t row_stride);
template <typename T>
int vad_frames(Ctx *c, const T *d_x, int64_t n_buf, int n_ch, int64_t base_sample, int64_t n_total, int sample_rate,
               double update_ms, int hop, int64_t t0, int64_t n_frames, unsigned char *d_vad);
template <typename T>
int rows_masked_sum(Ctx *c, const T *d_S, int64_t n_rows, int64_t row_len, int64_t row_stride, const unsigned char *d_mask,
                    double *d_sum, long long *d_count);
template <typename T>
int wiener_gain(Ctx *c, const double *d_noise_psd, const double *d_speech_psd, int64_t n_sig, int n_bins, int kernel_bins,
                T *d_gain);
template <typename T>
int stft_apply_gain(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len, int64_t max_len,
                    int n_ch, int n_fft, int hop, const T *d_gain, int64_t gain_stride, T *d_y, const int64_t *d_y_off,
                    int16_t *d_pcm16);

// ------------------------------------------------------------------ misc
template <typename T>
int window_rms(Ctx *c, const T *d_x, int64_t n_rows, int n_ch, int64_t win, double *d_rms);
template <typename T>
int pcm_encode(Ctx *c, const T *d_x, int64_t n, int bits, void *d_q, T *d_xq);
template <typename T>
int synth_noisy_speech(Ctx *c, T *d_x, int64_t n, int64_t start, int sample_rate, uint64_t seed,
                       uint64_t stream, double sigma, double speech_gain);

}  // namespace adn
