"""Wave-resident wavelet tiles against the LDS tiles alone (option ADN_OPT_ABLATE bit 4096 switches the
wave-resident tiles off): outputs and statistics must be bit-identical; then the time of both."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
SR = 48000
gen = torch.Generator(device="cuda").manual_seed(5)
ok = True
for wavelet, level, thr in (("db4", 2, "soft"), ("db4", 1, "hard"), ("db4", 3, "soft"), ("db4", 4, "soft"), ("db8", 1, "soft"),
                            ("db8", 2, "hard"), ("db8", 3, "soft"), ("db8", 4, "soft"), ("db8", 8, "soft"), ("db4", 6, "hard")):
    # blocks of different lengths: odd, just below / at / above one window, long
    lens = np.array([144000, 143999, 1023, 1024, 1025, 2047, 4096, 4097, 5000, 7681, 12345, 100001, 3, 64, 20000, 8192, 4160, 2048, 1040, 12352, 132300, 7168, 3584, 960], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens + 7)[:-1]]).astype(np.int64)
    total = int(off[-1] + lens[-1] + 7)
    x = torch.randn(total, dtype=torch.float32, device="cuda", generator=gen) * 0.1
    d_off, d_len = e._i64(off), e._i64(lens)
    res = []
    for ab in (0, 4096):
        e.lib.adn_set_option(e._h, 2, ab)
        y = torch.full_like(x, 7.0)
        st = e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, thr_mode=thr, want_stats=True,
                               max_frames=int(lens.max()))
        torch.cuda.synchronize()
        res.append((y.cpu().numpy(), [t.cpu().numpy() for t in st]))
    e.lib.adn_set_option(e._h, 2, 0)
    same_y = np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    same_st = all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    if not same_y:
        bad = np.nonzero(res[0][0].view(np.uint32) != res[1][0].view(np.uint32))[0]
        blk = np.searchsorted(off, bad[:8], side="right") - 1
        print("   first differences at", bad[:8], "blocks", blk, "in-block", bad[:8] - off[blk], "count", bad.size,
              res[0][0][bad[:4]], res[1][0][bad[:4]])
    if not same_st:
        for a, b in zip(res[0][1], res[1][1]):
            d = np.nonzero(a != b)[0]
            if d.size: print("   stats differ at blocks", d[:8], a[d[:4]], b[d[:4]])
    print(wavelet, level, thr, "outputs equal:", same_y, "statistics equal:", same_st, flush=True)
    ok = ok and same_y and same_st
print("ALL EQUAL" if ok else "MISMATCH")
if len(sys.argv) > 1:
    clips, n = 256, 30 * 48000
    x = torch.randn(clips * n, dtype=torch.float32, device="cuda") * 0.1
    bs = e.denoise_block_frames(SR, n)
    off = np.concatenate([c * n + np.arange(0, n, bs, dtype=np.int64) for c in range(clips)])
    d_off, d_len = e._i64(off), e._i64(np.full(off.shape, bs, dtype=np.int64))
    y = torch.empty_like(x)
    for wavelet, level in (("db4", 2), ("db8", 8)):
        for ab in (0, 4096):
            e.lib.adn_set_option(e._h, 2, ab)
            run = lambda: e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, max_frames=bs)
            run(); torch.cuda.synchronize()
            e.kernel_timing(True)
            for _ in range(3):
                run()
            kt = e.kernel_times()
            e.kernel_timing(False)
            tot = sum(v["total_ms"] for v in kt.values()) / 3
            print(wavelet, level, "ablate", ab, "total %.3f ms" % tot, {k[:28]: round(v["total_ms"] / 3, 3) for k, v in kt.items() if v["total_ms"] / 3 > 0.04}, flush=True)
    e.lib.adn_set_option(e._h, 2, 0)
