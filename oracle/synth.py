"""Deterministic synthetic "noisy speech" generator (TEST INFRASTRUCTURE).

Every value is produced with exactly-rounded IEEE-754 double operations only
(+, -, *, /, floor, abs, min, max, integer hashing).  No libm call is used, so
the numpy implementation here, the C implementation in ``oracle/c`` and the HIP
generator kernel (compiled with FP contraction off) agree bit for bit.

Signal model (per sample index i, sample rate sr, stream id s):
  noise  n = sigma * sqrt(3) * (u0+u1+u2+u3 - 2),  u_k = 16-bit slices of
         splitmix64(seed, s, i)  (Irwin-Hall approximation of a Gaussian)
  speech three harmonics of a gliding 140 Hz fundamental, amplitudes
         0.25 / 0.12 / 0.06, built from the parabolic sine approximation
         psin(p) = q * (8 - 16|q|), q = p - floor(p + 0.5), gated by a 3 Hz
         on/off syllable envelope with 10 ms linear ramps.
"""
import numpy as np

MASK64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15
SQRT3 = 1.7320508075688772  # correctly rounded double of sqrt(3)


def _splitmix64(z):
    """splitmix64 finaliser on a uint64 numpy array (wrap-around arithmetic)."""
    z = z.copy()
    with np.errstate(over="ignore"):
        z += np.uint64(GOLDEN)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


def hash_u64(seed, stream, idx):
    """64-bit hash of (seed, stream, index).  idx: uint64 array."""
    with np.errstate(over="ignore"):
        base = np.uint64((seed * GOLDEN + stream * 0xD1B54A32D192ED03) & MASK64)
        return _splitmix64(idx.astype(np.uint64) * np.uint64(0xA0761D6478BD642F) + base)


def unit_noise(seed, stream, idx):
    """Zero-mean, unit-variance Irwin-Hall(4) noise, exact in double."""
    h = hash_u64(seed, stream, idx)
    m = np.uint64(0xFFFF)
    s = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m)
         + ((h >> np.uint64(48)) & m)).astype(np.float64)
    # each slice is an integer in [0, 65535]; centre on the exact mean 2*65535
    return (s - 131070.0) * (SQRT3 / 65536.0)


def psin(p):
    """Parabolic approximation of sin(2*pi*p); exact IEEE basic ops only."""
    q = p - np.floor(p + 0.5)
    return q * (8.0 - 16.0 * np.abs(q))


def speech(idx, sr):
    t = idx.astype(np.float64) / float(sr)
    # fundamental 140 Hz with a slow +-20 Hz glide (0.5 Hz modulation)
    phase = 140.0 * t + (20.0 / 3.0) * psin(0.5 * t) * 0.5
    r = 3.0 * t
    r = r - np.floor(r)
    ramp = np.minimum(r, 0.5 - r) / 0.03
    env = np.minimum(np.maximum(ramp, 0.0), 1.0)
    s = 0.25 * psin(phase) + 0.12 * psin(2.0 * phase) + 0.06 * psin(3.0 * phase)
    return env * s


def noisy_speech(n, sr, seed=42, stream=0, sigma=0.05, start=0, speech_gain=1.0,
                 dtype=np.float64):
    """x[i] = speech_gain*speech(i) + sigma*noise(i) for i in [start, start+n)."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    x = speech_gain * speech(idx, sr) + sigma * unit_noise(seed, stream, idx)
    return x.astype(dtype)


def white_noise(n, seed=42, stream=1000, sigma=0.05, start=0, dtype=np.float64):
    idx = np.arange(start, start + n, dtype=np.uint64)
    return (sigma * unit_noise(seed, stream, idx)).astype(dtype)


def stereo_noisy_speech(n, sr, seed=42, stream=0, sigma=0.05, dtype=np.float64):
    """(n, 2) array: same speech in both channels (right at 0.8 gain), independent noise."""
    left = noisy_speech(n, sr, seed, 2 * stream, sigma)
    right = noisy_speech(n, sr, seed, 2 * stream + 1, sigma, speech_gain=0.8)
    return np.stack([left, right], axis=1).astype(dtype)


def bits_checksum(x):
    """Order-sensitive 64-bit checksum of the raw bits of a float array."""
    b = np.ascontiguousarray(x).view(np.uint64 if x.dtype == np.float64 else np.uint32)
    b = b.astype(np.uint64).ravel()
    with np.errstate(over="ignore"):
        w = _splitmix64(np.arange(b.size, dtype=np.uint64))
        return int(np.bitwise_xor.reduce(b * (w | np.uint64(1))))
