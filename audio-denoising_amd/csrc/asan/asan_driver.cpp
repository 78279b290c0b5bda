// Host-side sanitizer check of libadn_hip's C ABI (make asan): built with AddressSanitizer +
// UndefinedBehaviorSanitizer and linked against the sanitized host code of the library and a
// stand-in HIP runtime (hip_stub.cpp: host memory, no kernels).  Drives the argument checks, the
// error paths and the workspace / descriptor handling of every entry point; results of kernels are
// not looked at (there are none).  Exit code 0 and a last line "ASAN DRIVER OK" = clean.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../../include/adn.h"

extern "C" long adn_stub_launches(void);

static int g_fail = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, adn_last_error()); \
            ++g_fail;                                                       \
        }                                                                   \
    } while (0)

template <typename T>
static void stft_stages(adn_ctx *c, int n_sig, int64_t n, int n_ch) {
    // the stub's device memory is host memory: descriptors can be written directly
    std::vector<int64_t> off(n_sig), len(n_sig), yoff(n_sig);
    for (int s = 0; s < n_sig; ++s) {
        off[s] = (int64_t)s * n * n_ch;
        len[s] = n - 7 * s;
        yoff[s] = (int64_t)s * n;
    }
    std::vector<T> x((size_t)n_sig * n * n_ch, (T)0.01), y((size_t)n_sig * n), noise(1025, (T)0.001), base((size_t)n_sig * 1025),
        gain((size_t)n_sig * 1025);
    std::vector<int16_t> pcm((size_t)n_sig * n);
    const bool f32 = sizeof(T) == 4;
#define CALL(name, ...) (f32 ? name##_f32(__VA_ARGS__) : name##_f64(__VA_ARGS__))
    auto X = [&](auto *p) { return p; };
    (void)X;
    if constexpr (sizeof(T) == 4) {
        EXPECT(adn_noise_psd_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 2048, 512, 1, 95.0, 0.1, 5, 5, base.data()) == ADN_OK);
        EXPECT(adn_spectral_subtract_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, noise.data(), 0, 2048, 512, 2.2, 0.1, 3, 5,
                                         y.data(), yoff.data(), pcm.data()) == ADN_OK);
        EXPECT(adn_spectral_subtract_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, noise.data(), 0, 1024, 256, 2.2, 0.1, 5, 7,
                                         y.data(), yoff.data(), nullptr) == ADN_OK);
        EXPECT(adn_wiener_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 48000, 2048, 512, 20.0, 5, 1, y.data(), yoff.data(),
                              gain.data(), nullptr) == ADN_OK);
        EXPECT(adn_wiener_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 48000, 2048, 512, 10.0, 3, 0, y.data(), yoff.data(),
                              nullptr, pcm.data()) == ADN_OK);
        EXPECT(adn_wiener_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 48000, 512, 128, 20.0, 5, 1, y.data(), yoff.data(),
                              nullptr, nullptr) == ADN_OK);
        // error paths
        EXPECT(adn_spectral_subtract_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, noise.data(), 0, 2048, 500, 2.2, 0.1, 3, 5,
                                         y.data(), yoff.data(), nullptr) == ADN_ERR_UNSUPPORTED);                 // hop does not divide n_fft/2
        EXPECT(adn_spectral_subtract_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, noise.data(), 0, 1000, 250, 2.2, 0.1, 3, 5,
                                         y.data(), yoff.data(), nullptr) == ADN_ERR_ARG);                         // n_fft not a power of two
        EXPECT(adn_spectral_subtract_f32(c, x.data(), n_sig, off.data(), len.data(), 100, n_ch, noise.data(), 0, 2048, 512, 2.2, 0.1, 3, 5,
                                         y.data(), yoff.data(), nullptr) == ADN_ERR_SHORT);                       // window longer than the signal
        EXPECT(adn_wiener_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 48000, 2048, 512, 20.0, 4, 1, y.data(), yoff.data(),
                              nullptr, nullptr) == ADN_ERR_ARG);                                                  // even median kernel
        EXPECT(adn_wiener_f32(c, nullptr, n_sig, off.data(), len.data(), n, n_ch, 48000, 2048, 512, 20.0, 5, 1, y.data(), yoff.data(),
                              nullptr, nullptr) == ADN_ERR_ARG);
        EXPECT(adn_noise_psd_f32(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 2048, 512, 1, 101.0, 0.1, 5, 5, base.data()) == ADN_ERR_ARG);
    } else {
        EXPECT(adn_noise_psd_f64(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 2048, 512, 1, 95.0, 0.1, 5, 5, base.data()) == ADN_OK);
        EXPECT(adn_spectral_subtract_f64(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, noise.data(), 1025 * 0, 2048, 512, 2.2, 0.1, 3, 5,
                                         y.data(), yoff.data(), pcm.data()) == ADN_OK);
        EXPECT(adn_wiener_f64(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 44100, 2048, 512, 20.0, 5, 1, y.data(), yoff.data(),
                              gain.data(), nullptr) == ADN_OK);
        EXPECT(adn_wiener_f64(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, 44100, 4096, 1024, 30.0, 7, 0, y.data(), yoff.data(),
                              nullptr, nullptr) == ADN_OK);
        EXPECT(adn_spectral_subtract_f64(c, x.data(), n_sig, off.data(), len.data(), n, n_ch, noise.data(), 0, 4096, 1024, 2.2, 0.1, 3, 5,
                                         y.data(), yoff.data(), nullptr) == ADN_ERR_UNSUPPORTED);                 // does not fit LDS
    }
#undef CALL
}

template <typename T>
static void wavelet_blocks(adn_ctx *c, int n_blocks, int64_t frames, int n_ch, int wavelet, int level, int ext) {
    std::vector<int64_t> off(n_blocks), len(n_blocks), zeroed(n_blocks * (n_ch > 1 ? n_ch : 1));
    const int64_t stride = (frames + 2) * n_ch;
    for (int b = 0; b < n_blocks; ++b) {
        off[b] = b * stride;
        len[b] = frames - (b % 3);                          // odd and even lengths
    }
    std::vector<T> x((size_t)n_blocks * stride, (T)0.5), y((size_t)n_blocks * stride);
    std::vector<double> sigma(zeroed.size()), thr(zeroed.size());
    int rc;
    if constexpr (sizeof(T) == 4)
        rc = adn_wavelet_denoise_f32(c, x.data(), y.data(), n_blocks, off.data(), len.data(), off.data(), frames, n_ch,
                                     ADN_AXIS_REFERENCE, wavelet, level, ext, ADN_THR_SOFT, sigma.data(), thr.data(), zeroed.data());
    else
        rc = adn_wavelet_denoise_f64(c, x.data(), y.data(), n_blocks, off.data(), len.data(), off.data(), frames, n_ch,
                                     ADN_AXIS_REFERENCE, wavelet, level, ext, ADN_THR_HARD, sigma.data(), thr.data(), zeroed.data());
    EXPECT(rc == ADN_OK);
}

int main() {
    EXPECT(adn_abi_version() == ADN_ABI_VERSION);
    EXPECT(adn_stft_num_frames(2047, 2048, 512, 1) == -1);
    EXPECT(adn_stft_num_frames(2048, 2048, 512, 1) == 1);
    EXPECT(adn_istft_out_samples(160000, 2048, 512) == 158208);
    EXPECT(adn_wavelet_out_frames(16001) == 16002);
    EXPECT(adn_denoise_block_frames(16000, 160000) == 16000);
    EXPECT(adn_denoise_block_frames(0, 10) == 0);

    adn_ctx *c = nullptr;
    EXPECT(adn_create(nullptr, 0) == ADN_ERR_ARG);
    EXPECT(adn_create(&c, 3) != ADN_OK);                   // no such device
    setenv("ADN_STUB_ARCH", "gfx942:sramecc+:xnack-", 1);
    EXPECT(adn_create(&c, 0) != ADN_OK);                   // another architecture is refused
    setenv("ADN_STUB_ARCH", "gfx950:sramecc+:xnack-", 1);
    EXPECT(adn_create(&c, 0) == ADN_OK && c != nullptr);
    if (!c) return 1;
    EXPECT(adn_set_stream(c, nullptr) == ADN_OK);
    EXPECT(adn_set_option(c, 99, 1) != ADN_OK);
    EXPECT(adn_set_option(nullptr, ADN_OPT_FORCE_GENERIC, 1) != ADN_OK);

    // ---- wavelet blocks: every route's workspace arithmetic (single-pass statistics, deep levels, level by
    // level, symmetric extension, channel axis), growing and shrinking batches on one context
    wavelet_blocks<float>(c, 40, 144000, 1, ADN_WAVELET_DB8, 8, ADN_EXT_PERIODIZATION);
    wavelet_blocks<float>(c, 3, 2500001, 1, ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION);
    wavelet_blocks<float>(c, 11, 2137000, 2, ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION);
    wavelet_blocks<double>(c, 7, 16001, 1, ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION);
    wavelet_blocks<double>(c, 5, 400000, 2, ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION);
    wavelet_blocks<double>(c, 2, 4411, 1, ADN_WAVELET_DB8, 4, ADN_EXT_SYMMETRIC);
    wavelet_blocks<float>(c, 1, 7, 1, ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION);
    wavelet_blocks<float>(c, 2, 300, 4, ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION);
    EXPECT(adn_set_option(c, ADN_OPT_FORCE_GENERIC, 1) == ADN_OK);
    wavelet_blocks<float>(c, 6, 50000, 1, ADN_WAVELET_DB8, 8, ADN_EXT_PERIODIZATION);
    wavelet_blocks<float>(c, 6, 50000, 2, ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION);
    EXPECT(adn_set_option(c, ADN_OPT_FORCE_GENERIC, 0) == ADN_OK);
    {
        std::vector<int64_t> off{0}, len{1000};
        std::vector<float> x(3000), y(3000);
        EXPECT(adn_wavelet_denoise_f32(c, x.data(), y.data(), 1, off.data(), len.data(), off.data(), 1000, 3, ADN_AXIS_REFERENCE,
                                       ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION, ADN_THR_SOFT, nullptr, nullptr, nullptr) != ADN_OK);  // odd channel count
        EXPECT(adn_wavelet_denoise_f32(c, x.data(), y.data(), 1, off.data(), len.data(), off.data(), 1000, 1, ADN_AXIS_REFERENCE, 7, 2,
                                       ADN_EXT_PERIODIZATION, ADN_THR_SOFT, nullptr, nullptr, nullptr) != ADN_OK);                     // unknown wavelet
        EXPECT(adn_wavelet_denoise_f32(c, x.data(), y.data(), 1, off.data(), len.data(), off.data(), 1000, 1, ADN_AXIS_REFERENCE,
                                       ADN_WAVELET_DB4, 0, ADN_EXT_PERIODIZATION, ADN_THR_SOFT, nullptr, nullptr, nullptr) != ADN_OK);  // level 0
        EXPECT(adn_wavelet_denoise_f32(c, x.data(), nullptr, 1, off.data(), len.data(), off.data(), 1000, 1, ADN_AXIS_REFERENCE,
                                       ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION, ADN_THR_SOFT, nullptr, nullptr, nullptr) == ADN_ERR_ARG);
        // the entry point that allocates nothing: a small batch fits its fixed scratch, a large one is refused
        EXPECT(adn_wavelet_denoise_noalloc_f32(c, x.data(), y.data(), 1, off.data(), len.data(), off.data(), 1000, 1, ADN_AXIS_REFERENCE,
                                               ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION, ADN_THR_SOFT, nullptr, nullptr, nullptr) == ADN_OK);
        std::vector<int64_t> offs(4000), lens(4000, 288000);
        for (int b = 0; b < 4000; ++b) offs[b] = 0;
        EXPECT(adn_wavelet_denoise_noalloc_f32(c, x.data(), y.data(), 4000, offs.data(), lens.data(), offs.data(), 288000, 2,
                                               ADN_AXIS_REFERENCE, ADN_WAVELET_DB4, 2, ADN_EXT_PERIODIZATION, ADN_THR_SOFT, nullptr, nullptr,
                                               nullptr) == ADN_ERR_NOMEM);
    }

    // ---- STFT stages, both precisions, mono and stereo, short and long rows
    stft_stages<float>(c, 3, 60000, 1);
    stft_stages<float>(c, 2, 30000, 2);
    stft_stages<float>(c, 1, 6000000, 1);                  // rows of 11 715 frames: shared by several waves
    stft_stages<double>(c, 2, 50000, 2);
    stft_stages<double>(c, 1, 5000000, 1);

    // ---- segment statistics, wavedec / waverec (host-side band lengths), misc
    {
        std::vector<int64_t> off{0, 40000}, len{40000, 33};
        std::vector<double> x(80000, 1.0), med(2), mad(2);
        EXPECT(adn_mad_f64(c, x.data(), 2, off.data(), len.data(), 40000, med.data(), mad.data()) == ADN_OK);
        EXPECT(adn_mad_f64(c, x.data(), 2, off.data(), len.data(), 0, med.data(), mad.data()) == ADN_ERR_ARG);
        std::vector<double> co(41000), y(41000);
        int64_t bl[5] = {0}, n_out = 0;
        EXPECT(adn_wavedec_f64(c, x.data(), 4411, ADN_WAVELET_DB8, 4, ADN_EXT_SYMMETRIC, co.data(), (int64_t)co.size(), bl) == ADN_OK);
        EXPECT(bl[4] == (4411 + 15) / 2);
        EXPECT(adn_waverec_f64(c, co.data(), bl, ADN_WAVELET_DB8, 4, ADN_EXT_SYMMETRIC, y.data(), (int64_t)y.size(), &n_out) == ADN_OK);
        EXPECT(adn_wavedec_f64(c, x.data(), 4411, ADN_WAVELET_DB8, 4, ADN_EXT_SYMMETRIC, co.data(), 100, bl) != ADN_OK);       // capacity
        std::vector<double> rms(10);
        EXPECT(adn_window_rms_f64(c, x.data(), 40000, 2, 4410, rms.data()) == ADN_OK);
        std::vector<int16_t> q(1024);
        EXPECT(adn_pcm_encode_f64(c, x.data(), 1024, 16, q.data(), nullptr) == ADN_OK);
        EXPECT(adn_pcm_encode_f64(c, x.data(), 1024, 12, q.data(), nullptr) != ADN_OK);
    }

    // ---- kernel timing: a report into a buffer that is too small keeps the records
    EXPECT(adn_set_option(c, ADN_OPT_KERNEL_TIMING, 1) == ADN_OK);
    stft_stages<float>(c, 2, 30000, 1);
    {
        char small[8];
        const int64_t need = adn_kernel_times(c, small, sizeof(small));
        EXPECT(need < -1);
        std::vector<char> big((size_t)-need);
        EXPECT(adn_kernel_times(c, big.data(), (int64_t)big.size()) == -need - 1);
        EXPECT(strstr(big.data(), "k_apply_wave") != nullptr);
    }
    EXPECT(adn_set_option(c, ADN_OPT_KERNEL_TIMING, 0) == ADN_OK);

    // ---- allocation failure: a workspace that the (stub) runtime refuses
    setenv("ADN_STUB_MALLOC_LIMIT", "1000000", 1);
    {
        std::vector<int64_t> off{0}, len{30000000};
        std::vector<float> x(1), y(1);                      // never touched: the call fails before any launch
        const long before = adn_stub_launches();
        EXPECT(adn_wiener_f32(c, x.data(), 1, off.data(), len.data(), 30000000, 1, 48000, 2048, 512, 20.0, 5, 1, y.data(), off.data(), nullptr,
                              nullptr) == ADN_ERR_NOMEM);
        EXPECT(adn_stub_launches() == before);
        EXPECT(strstr(adn_last_error(), "allocation") != nullptr);
    }
    unsetenv("ADN_STUB_MALLOC_LIMIT");
    stft_stages<float>(c, 1, 40000, 1);                    // the context still works afterwards

    EXPECT(adn_synchronize(c) == ADN_OK);
    EXPECT(adn_destroy(c) == ADN_OK);
    EXPECT(adn_destroy(nullptr) == ADN_OK);                 // like free(NULL)
    if (g_fail) {
        fprintf(stderr, "%d expectation(s) failed\n", g_fail);
        return 1;
    }
    printf("launches enqueued on the stub: %ld\nASAN DRIVER OK\n", adn_stub_launches());
    return 0;
}
