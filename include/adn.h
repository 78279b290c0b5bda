/*
 * adn.h - C ABI of the MI355X-native audio denoising hot path (libadn_hip.so).
 *
 * The reference (nobbydoo80/Audio-Denoising) is pure Python and has no FFI of
 * its own: its hot path sits behind the class API in denoise.py and the four
 * stage scripts under src/.  This header is the boundary a ctypes binding in
 * those files would call instead of pywt / scipy / numpy.  Every entry point
 * names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *  - All functions return ADN_OK (0) or a negative ADN_ERR_* code; the message
 *    for the last failure on the calling thread is adn_last_error().
 *  - Pointers prefixed d_ are DEVICE pointers (hipMalloc / torch-ROCm tensors),
 *    pointers prefixed h_ are host pointers.  The caller owns every buffer.
 *  - Work is enqueued on the context's stream (adn_set_stream; default: the
 *    null stream) and is asynchronous with respect to the host; no entry point
 *    synchronises unless its comment says so.  Two one-time exceptions: the first STFT
 *    call of a context for a given (n_fft, precision) builds its window / twiddle tables
 *    with a blocking allocation and copy, and a call that has to grow the workspace
 *    waits for the stream before it frees the old one.
 *  - A context owns a device workspace that grows on demand (hipMalloc only on
 *    growth) and is re-used by later calls, so a context must not be used from
 *    two host threads at once.  Calls on one context are ordered by its stream.
 *  - _f32 entry points compute in IEEE binary32, _f64 in binary64 (the
 *    reference's own precision).  Order statistics are exact in either mode.  Where a
 *    binary32 kernel replaces a division or square root by the hardware's approximate
 *    reciprocal / square root (1 ulp: the soft threshold of mono wavelet blocks, the
 *    magnitudes and phases of the spectral subtraction), its result is not that of a
 *    correctly rounded binary32 restatement; the _f64 entry points use exact operations.
 */
#ifndef ADN_H
#define ADN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct adn_ctx adn_ctx;

enum {
    ADN_OK = 0,
    ADN_ERR_ARG = -1,          /* bad argument (null pointer, size, enum) */
    ADN_ERR_HIP = -2,          /* a HIP runtime call failed */
    ADN_ERR_UNSUPPORTED = -3,  /* valid in the reference but not implemented here */
    ADN_ERR_NOMEM = -4,        /* workspace allocation failed */
    ADN_ERR_SHORT = -5         /* signal shorter than the STFT window (scipy raises ValueError) */
};

enum { ADN_WAVELET_DB4 = 0, ADN_WAVELET_DB8 = 1 };
enum { ADN_EXT_PERIODIZATION = 0, ADN_EXT_SYMMETRIC = 1 };
enum { ADN_THR_SOFT = 0, ADN_THR_HARD = 1 };
/* Axis of the wavelet transform of multi-channel blocks.  ADN_AXIS_REFERENCE: what the reference
 * does (pywt's default axis=-1 on a (frames, channels) block: across the channels of each frame).
 * ADN_AXIS_TIME: every channel is denoised along time as a mono block of its own. */
enum { ADN_AXIS_REFERENCE = 0, ADN_AXIS_TIME = 1 };

/* ---------------------------------------------------------------- context */
/* ADN_ABI_VERSION counts incompatible changes of the argument lists below; a caller compares the value it was
 * built with against adn_abi_version() of the library it loaded before it calls anything else.
 *   2: adn_wavelet_denoise_* gained axis_mode and ext_mode (in the middle of the list: a caller built against the
 *      older header would pass wavelet / level / thr_mode in the wrong places); ADN_OPT_* renumbered.
 *   3: adn_kernel_times keeps its records when the buffer is too small. */
#define ADN_ABI_VERSION 3
int adn_abi_version(void);
const char *adn_last_error(void);
int adn_version(void);
int adn_create(adn_ctx **ctx, int device);
int adn_destroy(adn_ctx *ctx);
/* hip_stream is a hipStream_t (0 = null stream). */
int adn_set_stream(adn_ctx *ctx, void *hip_stream);
/* Blocks the host until everything enqueued on the context's stream is done. */
int adn_synchronize(adn_ctx *ctx);
/* Tuning / test knobs (value 0 restores normal operation).
 * ADN_OPT_FORCE_GENERIC: run the shape-generic kernels even where a specialised one exists
 *   (binary32, n_fft = 2048; fused wavelet tiles; single-pass threshold statistics), so that
 *   tests can cover both.
 * ADN_OPT_FORCE_FALLBACK: the single-pass threshold statistics hand every block to their exact
 *   per-block fallback (independent of ADN_OPT_FORCE_GENERIC).
 * ADN_OPT_ABLATE (diagnostics): a bit mask.
 *   Bits that make results WRONG (timing only): 1 analysis, 2 synthesis, 4 overlap-add of the shape-generic
 *   STFT kernels skipped (the mono tile kernels read 1, 2, 4, 8 the same way: levels, shrink, deeper levels,
 *   synthesis); 32 the row select decides no bit / the generic subtract kernel skips its magnitudes; 8192 the
 *   wave-resident wavelet tiles store no output.
 *   Bit 16: the fused STFT kernels record per-phase shader clocks of wave 0; the launch then ALLOCATES a counter
 *   buffer (hipMalloc), SYNCHRONISES the stream and prints the totals to stderr.
 *   Bits that change no result (they pick another route to the same values, for tests): 64 per-bin statistics on
 *   transposed rows (LDS select, separate voice-activity pass) instead of the grouped rows; 128 voice-activity
 *   energies in their own pass; 256 binary64 STFT stages on the shape-generic kernels; 512 mono blocks take their
 *   threshold statistics in two passes (finest band written to memory, two selections on it) instead of one pass
 *   over the samples; 1024 every grouped row is shared by two waves of the row select (what rows longer than
 *   8192 frames get by themselves); 2048 the single-pass statistics print the state of blocks that fell back to
 *   stderr; 4096 binary32 mono blocks run on the LDS tile kernels alone (no wave-resident tiles); 16384 power rows of
 *   any length take the streaming row select of the Wiener stage (sample bracket, one gathering pass, exact select on
 *   the candidates: what rows of more than 22 528 frames, binary64 24 576, take by themselves); 32768 no row does;
 *   65536 the rows the streaming select's brackets miss go to its radix selection (what rows of more than 90 112
 *   frames, binary64 49 152, get by themselves) instead of the register-resident select.
 * ADN_OPT_KERNEL_TIMING: record a HIP event pair on the context's stream around every kernel
 *   launch; adn_kernel_times() reports them.  Switching it off drops unreported records. */
enum { ADN_OPT_FORCE_GENERIC = 1, ADN_OPT_ABLATE = 2, ADN_OPT_FORCE_FALLBACK = 3, ADN_OPT_KERNEL_TIMING = 4 };
int adn_set_option(adn_ctx *ctx, int option, int value);
/* Measurement aid (no counterpart in the reference): synchronises the stream, then writes one
 * line per launch site timed since the last report, "name<TAB>launches<TAB>total_ms<NEWLINE>",
 * NUL-terminated, into buf and forgets the records.  Returns the text length, -1 on failure, or
 * minus the capacity needed when cap is too small (the records are kept: call again with that much room). */
int64_t adn_kernel_times(adn_ctx *ctx, char *buf, int64_t cap);
/* Measurement aid (no counterpart in the reference): while kernel timing is on, only launches of the
 * site called `name` (as adn_kernel_times() prints it) get their event pair; NULL or "" = every launch.
 * An event pair is a marker the GPU's command processor handles between two kernels: around every
 * launch of a chain of twenty kernels they cost a few percent, around one kernel nothing measurable. */
int adn_kernel_timing_only(adn_ctx *ctx, const char *name);
/* Bytes of device workspace currently held by the context. */
int64_t adn_workspace_bytes(adn_ctx *ctx);

/* ------------------------------------------------------- length helpers */
/* Frames pywt.waverec(mode='per') returns for a block of `frames` frames when
 * the transform runs along time (mono): 2*ceil(frames/2).  denoise.py:77,89. */
int64_t adn_wavelet_out_frames(int64_t frames);
/* Frames per block of AudioDeNoise.deNoise: int(rate * (frames/rate) * 0.10),
 * evaluated in IEEE double like the reference.  denoise.py:76. */
int64_t adn_denoise_block_frames(int64_t rate, int64_t frames);
/* scipy.signal.stft(boundary=None, padded=True) frame count for n samples, or
 * -1 when n < n_fft.  src/spectral_subtract.py:82. */
int64_t adn_stft_num_frames(int64_t n, int n_fft, int hop, int padded);
/* Samples scipy.signal.istft (boundary=True) returns: (T-1)*hop.
 * src/spectral_subtract.py:108. */
int64_t adn_istft_out_samples(int64_t n, int n_fft, int hop);

/* ------------------------------------------------ phase 2: wavelet blocks
 * Replaces, per block, pywt.wavedec(block, w, mode='per', level=L) ->
 * sigma = mad(cD1) -> thresh = sigma*sqrt(2*ln(frames)) -> pywt.threshold on
 * every detail band -> pywt.waverec  (denoise.py:16-23, 77-89).
 *
 * Blocks are independent.  Block b reads frames[b] frames of n_ch interleaved
 * samples starting at element d_in[d_in_off[b]] and writes to
 * d_out[d_out_off[b]] (element offsets).
 *   n_ch == 1 : the transform runs along time; the block comes back with
 *               adn_wavelet_out_frames(frames[b]) samples.
 *   n_ch >= 2, axis_mode ADN_AXIS_REFERENCE : like the reference (pywt default
 *               axis=-1 on a (frames, n_ch) block) the transform runs ACROSS THE
 *               CHANNEL AXIS of every frame; n_ch must be even and <= 8,
 *               periodization only; output is frames[b]*n_ch.
 *   n_ch >= 2, axis_mode ADN_AXIS_TIME : every channel is a mono block of its own
 *               (own sigma and threshold); output is
 *               adn_wavelet_out_frames(frames[b]) frames of n_ch samples.
 * ext_mode: ADN_EXT_PERIODIZATION (the reference's mode='per') or
 * ADN_EXT_SYMMETRIC (pywt's 'symmetric', level-by-level kernels).
 * d_in_off / d_frames / d_out_off: device arrays of n_blocks int64.
 * max_frames: an upper bound of frames[b] (sizes grids and workspace).
 * d_sigma / d_thresh (doubles) and d_zeroed (int64: detail coefficients whose
 * magnitude was below the threshold) may be NULL; they hold one value per block,
 * or per block and channel ([b * n_ch + ch]) under ADN_AXIS_TIME with n_ch >= 2.
 */
int adn_wavelet_denoise_f32(adn_ctx *ctx, const float *d_in, float *d_out, int64_t n_blocks,
                            const int64_t *d_in_off, const int64_t *d_frames,
                            const int64_t *d_out_off, int64_t max_frames, int n_ch, int axis_mode,
                            int wavelet, int level, int ext_mode, int thr_mode, double *d_sigma,
                            double *d_thresh, int64_t *d_zeroed);
int adn_wavelet_denoise_f64(adn_ctx *ctx, const double *d_in, double *d_out, int64_t n_blocks,
                            const int64_t *d_in_off, const int64_t *d_frames,
                            const int64_t *d_out_off, int64_t max_frames, int n_ch, int axis_mode,
                            int wavelet, int level, int ext_mode, int thr_mode, double *d_sigma,
                            double *d_thresh, int64_t *d_zeroed);

/* adn_wavelet_denoise_f32 for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, same kernels, same results; the kernels' workspace (candidate
 * lists and counters of the threshold statistics, finest-band coefficients of transforms deeper
 * than four levels) lives in a static
 * device array of ADN_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the
 * process ends.  The call makes no hipMalloc / hipFree and no other allocating call; a batch whose
 * kernels need more scratch than that is rejected with ADN_ERR_NOMEM (the message states both
 * byte counts), and so is ADN_AXIS_TIME with n_ch >= 2 (its planar copies need a second buffer).
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two
 * contexts, two streams) race; calls on one stream are safe.  Every call initialises what it
 * reads, so nothing of an earlier call's contents leaks into a later one.  Kernel timing
 * (ADN_OPT_KERNEL_TIMING) is not recorded for this entry point. */
#define ADN_NOALLOC_SCRATCH_BYTES (32u << 20)
int adn_wavelet_denoise_noalloc_f32(adn_ctx *ctx, const float *d_in, float *d_out, int64_t n_blocks,
                                    const int64_t *d_in_off, const int64_t *d_frames,
                                    const int64_t *d_out_off, int64_t max_frames, int n_ch,
                                    int axis_mode, int wavelet, int level, int ext_mode, int thr_mode,
                                    double *d_sigma, double *d_thresh, int64_t *d_zeroed);

/* mad() of denoise.py:16-23 for a batch of segments: d_mad[s] = median(|x - median(x)|)
 * over the len[s] values at d_x[d_off[s]] (numpy semantics: even counts average the two
 * middle order statistics; no 0.6745 factor).  d_median (may be NULL) receives the medians.
 * d_off / d_len: device arrays of n_seg int64; max_len bounds len[s]. */
int adn_mad_f32(adn_ctx *ctx, const float *d_x, int64_t n_seg, const int64_t *d_off,
                const int64_t *d_len, int64_t max_len, double *d_median, double *d_mad);
int adn_mad_f64(adn_ctx *ctx, const double *d_x, int64_t n_seg, const int64_t *d_off,
                const int64_t *d_len, int64_t max_len, double *d_median, double *d_mad);

/* Multi-level analysis / synthesis of ONE signal (pywt.wavedec / pywt.waverec,
 * and the single-level steps pywt.WaveletPacket is built from:
 * lib/windowBundle.py:33-34).  d_coeffs receives the bands concatenated as
 * [cA_L, cD_L, ..., cD_1]; h_band_len (level+1 int64, host) receives their
 * lengths.  ext_mode: ADN_EXT_PERIODIZATION or ADN_EXT_SYMMETRIC.
 * These two calls synchronise the stream before returning. */
int adn_wavedec_f64(adn_ctx *ctx, const double *d_x, int64_t n, int wavelet, int level,
                    int ext_mode, double *d_coeffs, int64_t coeffs_capacity,
                    int64_t *h_band_len);
int adn_waverec_f64(adn_ctx *ctx, const double *d_coeffs, const int64_t *h_band_len, int wavelet,
                    int level, int ext_mode, double *d_y, int64_t y_capacity, int64_t *h_n_out);
int adn_wavedec_f32(adn_ctx *ctx, const float *d_x, int64_t n, int wavelet, int level,
                    int ext_mode, float *d_coeffs, int64_t coeffs_capacity, int64_t *h_band_len);
int adn_waverec_f32(adn_ctx *ctx, const float *d_coeffs, const int64_t *h_band_len, int wavelet,
                    int level, int ext_mode, float *d_y, int64_t y_capacity, int64_t *h_n_out);

/* --------------------------------------------------- STFT stages (batched)
 * Signals are independent.  Signal s has len[s] frames of n_ch interleaved
 * samples at d_x[d_off[s]]; multi-channel input is folded to mono with
 * mean(axis=1) like the reference (src/noise_profile.py:45,
 * src/spectral_subtract.py:76, src/wiener_filter.py:67).  d_off / d_len are
 * device arrays of n_sig int64; max_len bounds len[s].  Every signal must have
 * len >= n_fft (else ADN_ERR_SHORT is reported by the host wrapper, which knows
 * the lengths; the kernels clamp).  n_fft must be a power of two in [64, 4096].
 * Any element offsets are accepted, but the fused kernels load a signal's frames
 * in 16-byte pieces and store its output in 8-byte pieces: signals (and outputs,
 * d_y_off) that start at 16-byte aligned addresses take those paths, others a
 * slower element-wise one (spectral subtraction of files packed back to back at
 * odd offsets: 1.7 times the time).  The same holds for the blocks of
 * adn_wavelet_denoise_* (d_in_off / d_out_off).  Files of different lengths
 * may share a call: the fused kernels (n_fft 2048, hop 512, up to 4096 signals)
 * size their runs on the device from all lengths (20 ... 60 s files in one
 * call: about 1.05 times the time of equal files with the same total audio).
 */

/* Phase 1 (src/noise_profile.py:11-14, 57-63): |STFT| with a periodic Hann
 * window of n_fft samples, P-th percentile over time per bin (linear
 * interpolation), clamp to spectral_floor * max|Z|, box filter of freq_bins
 * along frequency with edge replication.  d_baseline receives n_sig rows of
 * n_fft/2+1 values: column 0 of the reference's time-constant baseline_mag.
 * padded = the config's `center` flag, which the reference hands to scipy as
 * `padded`. */
int adn_noise_psd_f32(adn_ctx *ctx, const float *d_x, int64_t n_sig, const int64_t *d_off,
                      const int64_t *d_len, int64_t max_len, int n_ch, int n_fft, int hop,
                      int padded, double percentile, double spectral_floor, int freq_bins,
                      int time_frames, float *d_baseline);
int adn_noise_psd_f64(adn_ctx *ctx, const double *d_x, int64_t n_sig, const int64_t *d_off,
                      const int64_t *d_len, int64_t max_len, int n_ch, int n_fft, int hop,
                      int padded, double percentile, double spectral_floor, int freq_bins,
                      int time_frames, double *d_baseline);

/* Phase-1 analysis summaries (src/analyze_audio.py:24-26 compute_spectrogram, :50-62
 * summarize_band_energy, :161 noise_profile_time_mean_mag): |STFT| as above, then for signal s
 * and band b (the bins k with h_lo_hz[b] <= k * sample_rate / n_fft <= h_hi_hz[b], both ends
 * inclusive, frequencies built like scipy's rfftfreq):
 *   d_mean_mag[s * n_bands + b]                    = mean of |Z| over the band's bins and all frames
 *   d_perc_db[(p * n_sig + s) * n_bands + b]       = h_perc[p]-th percentile (linear interpolation) of
 *                                                    20*log10(|Z| + 1e-12) over the same values
 *   d_time_mean[s * T_max + t]                     = mean over all bins of |Z[:, t]|  (NULL: skipped)
 *   d_mag[(s * (n_fft/2+1) + k) * T_max + t]       = |Z[k, t]| itself, for the host-side plots (NULL: skipped)
 * with T_max = adn_stft_num_frames(max_len, n_fft, hop, padded); entries at frames t beyond a shorter signal's own
 * frame count are unspecified in d_time_mean and d_mag.  At most 16 bands and 16
 * percentiles per call; an empty band gives NaN.  h_* arrays are host memory. */
int adn_band_summary_f32(adn_ctx *ctx, const float *d_x, int64_t n_sig, const int64_t *d_off,
                         const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate, int n_fft,
                         int hop, int padded, int n_bands, const double *h_lo_hz,
                         const double *h_hi_hz, int n_perc, const double *h_perc, double *d_mean_mag,
                         double *d_perc_db, float *d_time_mean, float *d_mag);
int adn_band_summary_f64(adn_ctx *ctx, const double *d_x, int64_t n_sig, const int64_t *d_off,
                         const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate, int n_fft,
                         int hop, int padded, int n_bands, const double *h_lo_hz,
                         const double *h_hi_hz, int n_perc, const double *h_perc, double *d_mean_mag,
                         double *d_perc_db, double *d_time_mean, double *d_mag);

/* Phase 3, harmonic-percussive separation (src/hpss_separate.py:41-67 median_filtering_hpss,
 * :86-113 multiband_hpss, :115-140 apply_masks, :142-190 preserve_transients, :233-283 main):
 * |STFT| (n_fft window, padded) -> median over kernel_size frames (harmonic) and over kernel_size
 * bins (percussive), edges reflected like scipy.ndimage mode='reflect' -> masks
 * ((S*margin)^power, normalised with 1e-10; ADN_MASK_HARD: winner takes all; ADN_MASK_BINARY:
 * > 0.5) -> ISTFT of Z*mask_h and Z*mask_p (not clipped) -> optional transient preservation
 * (frames of 2048 / hop 512 whose original RMS exceeds the harmonic RMS by more than
 * threshold * max(original RMS) are blended: blend*x + (1-blend)*y_h, x NOT shifted by the
 * n_fft/2 samples the ISTFT drops, as in the reference) -> both outputs scaled by 0.95 / max|y|
 * when that maximum exceeds 0.95.
 * n_bands > 0: multiband mode, the bins of band b (h_lo_hz[b] <= f <= h_hi_hz[b]) are filtered
 * with the reflection at the band's own edges, a bin listed in several bands belongs to the
 * last one, bins in no band get zero masks.  kernel_size odd, <= 31.
 * d_yh / d_yp: outputs of adn_istft_out_samples(len[s], n_fft, hop) samples at d_y_off[s].
 * d_stats (may be NULL): 5 doubles per signal: harmonic RMS, percussive RMS, RMS of the mono
 * input, max|y| before scaling, number of transient frames.
 * Synchronises: every call uploads its per-bin band table from host memory and waits for the
 * stream (hipStreamSynchronize) before it enqueues its kernels. */
enum { ADN_MASK_SOFT = 0, ADN_MASK_HARD = 1, ADN_MASK_BINARY = 2 };
int adn_hpss_f32(adn_ctx *ctx, const float *d_x, int64_t n_sig, const int64_t *d_off,
                 const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate, int n_fft, int hop,
                 int kernel_size, double margin_h, double margin_p, double power, int mask_type,
                 int n_bands, const double *h_lo_hz, const double *h_hi_hz, int preserve_transients,
                 double threshold, double blend, float *d_yh, float *d_yp, const int64_t *d_y_off,
                 double *d_stats);
int adn_hpss_f64(adn_ctx *ctx, const double *d_x, int64_t n_sig, const int64_t *d_off,
                 const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate, int n_fft, int hop,
                 int kernel_size, double margin_h, double margin_p, double power, int mask_type,
                 int n_bands, const double *h_lo_hz, const double *h_hi_hz, int preserve_transients,
                 double threshold, double blend, double *d_yh, double *d_yp, const int64_t *d_y_off,
                 double *d_stats);

/* Phase 2b (src/spectral_subtract.py:29-46, 79-109): STFT -> max(|Z|-alpha*n,
 * beta*|Z|) -> box filter (smooth_freq x smooth_time, edge replicated) -> keep
 * the phase -> ISTFT (drops n_fft/2 at both ends) -> clip to [-1, 1].
 * d_noise: noise magnitude per bin, n_fft/2+1 values per signal at
 * d_noise[s*noise_stride] (noise_stride = 0 shares one profile).
 * Signal s writes adn_istft_out_samples(len[s]) samples at d_y[d_y_off[s]].
 * d_pcm16 (may be NULL): the stage file of the reference is 16-bit PCM
 * (src/spectral_subtract.py:109-110); when given, the kernel's store also writes
 * the file's sample values there (same offsets as d_y, rule of adn_pcm_encode_*)
 * and d_y holds what a reader of that file gets, pcm / 32768. */
int adn_spectral_subtract_f32(adn_ctx *ctx, const float *d_x, int64_t n_sig, const int64_t *d_off,
                              const int64_t *d_len, int64_t max_len, int n_ch,
                              const float *d_noise, int64_t noise_stride, int n_fft, int hop,
                              double alpha, double beta, int smooth_time, int smooth_freq,
                              float *d_y, const int64_t *d_y_off, int16_t *d_pcm16);
int adn_spectral_subtract_f64(adn_ctx *ctx, const double *d_x, int64_t n_sig,
                              const int64_t *d_off, const int64_t *d_len, int64_t max_len,
                              int n_ch, const double *d_noise, int64_t noise_stride, int n_fft,
                              int hop, double alpha, double beta, int smooth_time,
                              int smooth_freq, double *d_y, const int64_t *d_y_off,
                              int16_t *d_pcm16);

/* Phase 2c (src/wiener_filter.py:12-39, 72-129): |Z|^2, 10th percentile over
 * time per bin -> medfilt -> floor 1e-12; energy VAD (frame = int(sr*update_ms/
 * 1000) samples, -40 dB, hangover 5) mapped onto STFT frames; VAD-masked mean
 * (or the median when vad_guided == 0) -> medfilt -> max(., noise); gain =
 * medfilt(clip(snr/(1+snr), 0, 1)); Z*gain -> ISTFT -> clip.  The reference's
 * attack/release loop never executes, so the gain is constant in time.
 * d_gain (n_sig rows of n_fft/2+1, may be NULL) receives the per-bin gain.
 * d_pcm16 (may be NULL): as for adn_spectral_subtract_*. */
int adn_wiener_f32(adn_ctx *ctx, const float *d_x, int64_t n_sig, const int64_t *d_off,
                   const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate, int n_fft,
                   int hop, double update_ms, int kernel_bins, int vad_guided, float *d_y,
                   const int64_t *d_y_off, float *d_gain, int16_t *d_pcm16);
int adn_wiener_f64(adn_ctx *ctx, const double *d_x, int64_t n_sig, const int64_t *d_off,
                   const int64_t *d_len, int64_t max_len, int n_ch, int sample_rate, int n_fft,
                   int hop, double update_ms, int kernel_bins, int vad_guided, double *d_y,
                   const int64_t *d_y_off, double *d_gain, int16_t *d_pcm16);

/* ------------------------------------- one recording split over several devices
 * The reference loads a whole file into one spectrogram (src/spectral_subtract.py:75,82,
 * src/wiener_filter.py:66,72) and denoises it in blocks of 10 % of the file (denoise.py:76); a
 * recording too long for one device is cut into time chunks instead.  Everything that is local in
 * time runs on a chunk plus a halo of its neighbours' samples through the ordinary entry points
 * (the caller drops the halo's output); the entry points below are the pieces whose statistics
 * span the whole block / file: they expose the partial results that the holders of the chunks add
 * up (an all-reduce, or a loop over chunks on one device) before every holder finishes alone.
 *
 * Distributed order statistic of segments whose values are scattered over the holders (per-block
 * MAD of denoise.py:16-23,80; per-bin percentile of src/wiener_filter.py:84): an 8-bit radix descent
 * on the order-preserving integer keys.  Shared state per segment: d_prefix (the key bits decided so
 * far, zero at the start) and d_rank (0-based rank still to be found below the prefix, initially the
 * wanted rank among ALL values of the segment).  Round r (done_bits = 8 r): every holder calls
 * adn_select_hist_* on its part (d_hist: n_seg x 256 counters, cleared by the call), the histograms
 * are summed over the holders, every holder calls adn_select_resolve on the sum.  After 4 rounds
 * (binary32) / 8 rounds (binary64) adn_select_value_* turns the prefix into the value.  absdev != 0:
 * the values are |x - d_center[seg]| (second median of the MAD).  A NaN sorts above every number. */
int adn_select_hist_f32(adn_ctx *ctx, const float *d_vals, int64_t n_seg, const int64_t *d_off,
                        const int64_t *d_len, int64_t max_len, int absdev, const double *d_center,
                        const uint64_t *d_prefix, int done_bits, uint32_t *d_hist);
int adn_select_hist_f64(adn_ctx *ctx, const double *d_vals, int64_t n_seg, const int64_t *d_off,
                        const int64_t *d_len, int64_t max_len, int absdev, const double *d_center,
                        const uint64_t *d_prefix, int done_bits, uint32_t *d_hist);
int adn_select_resolve(adn_ctx *ctx, const uint32_t *d_hist, int64_t n_seg, int key_bits, int done_bits,
                       int64_t *d_rank, uint64_t *d_prefix);
int adn_select_value_f32(adn_ctx *ctx, const uint64_t *d_prefix, int64_t n_seg, double *d_out);
int adn_select_value_f64(adn_ctx *ctx, const uint64_t *d_prefix, int64_t n_seg, double *d_out);
/* The statistic from the two order statistics that bracket it, rounded as the one-call entry points
 * round it (numpy computes in the array's dtype).  n_values = size of a whole segment.  q < 0: the
 * median (ranks (n-1)/2 and n/2: numpy.median, denoise.py:21-22); 0 <= q <= 100: numpy.percentile's
 * linear interpolation between ranks floor((n-1) q/100) and that + 1 (src/wiener_filter.py:84). */
int adn_select_combine_f32(adn_ctx *ctx, const double *d_lo, const double *d_hi, int64_t n_seg,
                           int64_t n_values, double q, double *d_out);
int adn_select_combine_f64(adn_ctx *ctx, const double *d_lo, const double *d_hi, int64_t n_seg,
                           int64_t n_values, double q, double *d_out);

/* adn_wavelet_denoise_* for mono blocks with the thresholds handed in (d_thresh: one per block)
 * instead of computed from the block: pywt.wavedec -> pywt.threshold(value = d_thresh[b]) ->
 * pywt.waverec (denoise.py:77,86,89).  A holder of a chunk of a block runs it on the chunk plus a
 * halo (a multiple of 2^level samples on both sides, at least 2 (taps - 1)(2^level - 1) of them; the
 * halo beyond a block's end is the block's other end: periodization) and keeps the chunk's samples. */
int adn_wavelet_shrink_f32(adn_ctx *ctx, const float *d_in, float *d_out, int64_t n_blocks,
                           const int64_t *d_in_off, const int64_t *d_frames, const int64_t *d_out_off,
                           int64_t max_frames, int wavelet, int level, int ext_mode, int thr_mode,
                           const double *d_thresh, int64_t *d_zeroed);
int adn_wavelet_shrink_f64(adn_ctx *ctx, const double *d_in, double *d_out, int64_t n_blocks,
                           const int64_t *d_in_off, const int64_t *d_frames, const int64_t *d_out_off,
                           int64_t max_frames, int wavelet, int level, int ext_mode, int thr_mode,
                           const double *d_thresh, int64_t *d_zeroed);

/* |Z| (power = 0) or |Z|^2 (power = 1) of every STFT frame (scipy.signal.stft, boundary=None:
 * src/noise_profile.py:11-14, src/wiener_filter.py:72-75), transposed for row statistics:
 * d_S[(s * (n_fft/2+1) + k) * row_stride + t]; row_stride a multiple of 32, >= the frame count. */
int adn_stft_rows_f32(adn_ctx *ctx, const float *d_x, int64_t n_sig, const int64_t *d_off,
                      const int64_t *d_len, int64_t max_len, int n_ch, int n_fft, int hop, int padded,
                      int power, float *d_S, int64_t row_stride);
int adn_stft_rows_f64(adn_ctx *ctx, const double *d_x, int64_t n_sig, const int64_t *d_off,
                      const int64_t *d_len, int64_t max_len, int n_ch, int n_fft, int hop, int padded,
                      int power, double *d_S, int64_t row_stride);
/* Energy-VAD flags (src/wiener_filter.py:12-33, 92-98) of STFT frames t0 .. t0 + n_frames - 1 of ONE
 * signal of n_total samples, from a buffer of n_buf samples that starts at sample base_sample of the
 * signal.  The buffer must reach back five VAD frames (hangover) before the VAD frame of STFT frame
 * t0 and forward to the end of the VAD frame of the last STFT frame (or the signal's end). */
int adn_vad_frames_f32(adn_ctx *ctx, const float *d_x, int64_t n_buf, int n_ch, int64_t base_sample,
                       int64_t n_total, int sample_rate, double update_ms, int hop, int64_t t0,
                       int64_t n_frames, uint8_t *d_vad);
int adn_vad_frames_f64(adn_ctx *ctx, const double *d_x, int64_t n_buf, int n_ch, int64_t base_sample,
                       int64_t n_total, int sample_rate, double update_ms, int hop, int64_t t0,
                       int64_t n_frames, uint8_t *d_vad);
/* d_sum[row] = sum over t < row_len with d_mask[t] != 0 of d_S[row * row_stride + t] (binary64);
 * *d_count = number of such t.  (Numerator and denominator of src/wiener_filter.py:99-102.) */
int adn_rows_masked_sum_f32(adn_ctx *ctx, const float *d_S, int64_t n_rows, int64_t row_len,
                            int64_t row_stride, const uint8_t *d_mask, double *d_sum, int64_t *d_count);
int adn_rows_masked_sum_f64(adn_ctx *ctx, const double *d_S, int64_t n_rows, int64_t row_len,
                            int64_t row_stride, const uint8_t *d_mask, double *d_sum, int64_t *d_count);
/* Per-bin gain from the noise PSD (10th percentile) and the speech PSD (masked mean or median), both
 * before their running medians: src/wiener_filter.py:85-86, 108-115, 35-39.  n_sig rows of n_bins. */
int adn_wiener_gain_f32(adn_ctx *ctx, const double *d_noise_psd, const double *d_speech_psd, int64_t n_sig,
                        int n_bins, int kernel_bins, float *d_gain);
int adn_wiener_gain_f64(adn_ctx *ctx, const double *d_noise_psd, const double *d_speech_psd, int64_t n_sig,
                        int n_bins, int kernel_bins, double *d_gain);
/* STFT -> Z * gain -> ISTFT -> clip (src/wiener_filter.py:127-129) with the gain rows handed in:
 * d_gain[s * gain_stride + k] (gain_stride = 0 shares one row).  Other arguments as adn_wiener_*. */
int adn_stft_apply_gain_f32(adn_ctx *ctx, const float *d_x, int64_t n_sig, const int64_t *d_off,
                            const int64_t *d_len, int64_t max_len, int n_ch, int n_fft, int hop,
                            const float *d_gain, int64_t gain_stride, float *d_y, const int64_t *d_y_off,
                            int16_t *d_pcm16);
int adn_stft_apply_gain_f64(adn_ctx *ctx, const double *d_x, int64_t n_sig, const int64_t *d_off,
                            const int64_t *d_len, int64_t max_len, int n_ch, int n_fft, int hop,
                            const double *d_gain, int64_t gain_stride, double *d_y, const int64_t *d_y_off,
                            int16_t *d_pcm16);

/* ------------------------------------------------- RMS-window profiler
 * Per-window sum of squares over all channels divided by the window's row
 * count, then sqrt (lib/windowBundle.py:67-74 as used by
 * lib/noiseProfiler.py:151-162,191-206).  Window w covers rows
 * [w*win, min((w+1)*win, n_rows)).  d_rms receives ceil(n_rows/win) doubles. */
int adn_window_rms_f32(adn_ctx *ctx, const float *d_x, int64_t n_rows, int n_ch, int64_t win,
                       double *d_rms);
int adn_window_rms_f64(adn_ctx *ctx, const double *d_x, int64_t n_rows, int n_ch, int64_t win,
                       double *d_rms);

/* ---------------------------------------------------- PCM file samples
 * The stage files between the phases are 16-bit PCM (denoise.py:75,90; sf.write in
 * src/wavelet_denoise.py, src/spectral_subtract.py:109-110, src/wiener_filter.py): the next
 * stage reads quantised samples.  d_q receives the integer sample values of d_x[0..n)
 * (bits = 16: int16_t; bits = 24 or 32: int32_t) by libsndfile's rule with clipping enabled:
 * rint(x * 2^31) saturated to int32, arithmetic shift right by 32 - bits; NaN -> 0.
 * d_xq (may be null, may be d_x itself) receives q / 2^(bits-1), the value a reader of the
 * file gets, so that stages can be chained on the device while the files are still written
 * from d_q.  All three buffers must be 16-byte aligned. */
int adn_pcm_encode_f32(adn_ctx *ctx, const float *d_x, int64_t n, int bits, void *d_q, float *d_xq);
int adn_pcm_encode_f64(adn_ctx *ctx, const double *d_x, int64_t n, int bits, void *d_q, double *d_xq);

/* ------------------------------------------------ synthetic test signal
 * Fills d_x with n samples of the bit-reproducible "noisy speech" generator
 * used by the benchmark and the tests (splitmix64 hash -> Irwin-Hall noise +
 * parabolic-sine harmonics; only exactly rounded IEEE operations, so the host
 * generator produces the same bits).  No counterpart in the reference, which
 * ships no test signal.  Sample i of the call is generator index start+i. */
int adn_synth_noisy_speech_f64(adn_ctx *ctx, double *d_x, int64_t n, int64_t start,
                               int sample_rate, uint64_t seed, uint64_t stream, double sigma,
                               double speech_gain);
int adn_synth_noisy_speech_f32(adn_ctx *ctx, float *d_x, int64_t n, int64_t start,
                               int sample_rate, uint64_t seed, uint64_t stream, double sigma,
                               double speech_gain);

#ifdef __cplusplus
}
#endif
#endif /* ADN_H */
