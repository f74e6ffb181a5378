"""CPU, exhaustive: the integer identities the fused downscale kernel's pixel path rests on (psd_resize_kernels.hip, round 6).

OpenCV's 8-bit bilinear resize (resize.cpp, HResizeLinear / VResizeLinear with 11-bit coefficients) computes per channel
    h  = p0 * a0 + p1 * a1                       a0 + a1 = 2048, p in 0..255          (horizontal pass, per tap row)
    c  = ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2               (vertical pass)
The kernel computes hx = dot2((p0, p1), (a0 << 4, a1 << 4)) = h << 4 in one v_dot2_u32_u16, x = hx & 0xffff00, and each vertical product
as the upper 32 bits of the 24-bit multiply x * (b << 8) (v_mul_hi_u32_u24).  Here: every value the operands can take."""
import numpy as np


def test_masked_dot_product_is_the_truncated_horizontal_sum():
    h = np.arange(0, 255 * 2048 + 1, dtype=np.uint64)                 # every horizontal sum there is
    hx = h << np.uint64(4)
    assert int(hx.max()) < 1 << 23                                     # a 24-bit factor with room to spare
    x = hx & np.uint64(0x00FFFF00)
    assert np.array_equal(x, (h >> np.uint64(4)) << np.uint64(8))
    # the coefficient halves: a << 4 for a in 0 .. 2048 fits the 16 bits the dot product reads, and the sum fits 32
    assert (2048 << 4) < 1 << 16 and 255 * (2048 << 4) * 2 < 1 << 32


def test_upper_half_of_the_24_bit_product_is_the_vertical_term():
    q = np.arange(0, (255 * 2048 >> 4) + 1, dtype=np.uint64)           # h >> 4: 0 .. 32640
    x = q << np.uint64(8)
    assert int(x.max()) < 1 << 24
    for b0 in range(0, 2049, 64):                                      # all 2049 coefficients, 64 at a time
        b = np.arange(b0, min(b0 + 64, 2049), dtype=np.uint64)
        y = b << np.uint64(8)
        assert int(y.max()) < 1 << 24
        hi = (x[:, None] * y[None, :]) >> np.uint64(32)                # v_mul_hi_u32_u24
        assert np.array_equal(hi, (q[:, None] * b[None, :]) >> np.uint64(16))


def test_result_is_a_byte_and_dead_slots_are_zero():
    # the largest value: both rows 255 everywhere, b0 + b1 = 2048
    q = 255 * 2048 >> 4
    for b0 in (0, 1, 1024, 2047, 2048):
        c = (((b0 * q) >> 16) + (((2048 - b0) * q) >> 16) + 2) >> 2
        assert 0 <= c <= 255
    # a slot with all coefficients zero yields pixel 0 -> luma bin 0, S = 0, H = 0: nothing for the sums
    assert ((0 + 0 + 2) >> 2) == 0 and ((0 * 1868 + 0 * 9617 + 0 * 4899 + 8192) >> 14) == 0


def test_hue_wrap_by_unsigned_minimum():
    hh = np.arange(-90, 180, dtype=np.int64)                           # every value of the rounded hue product
    wrapped = np.minimum(hh.astype(np.uint32), (hh + 180).astype(np.uint32))
    assert np.array_equal(wrapped.astype(np.int64), np.where(hh < 0, hh + 180, hh))


def test_tap_bytes_out_of_three_aligned_dwords():
    """v_alignbyte_b32 twice puts the six tap bytes at byte 0 for every byte shift; v_perm_b32 with the kernel's selectors pairs the
    channels (emulated: the semantics the kernel relies on, the hardware itself is covered by the GPU parity tests)."""
    rng = np.random.default_rng(1)

    def alignbyte(hi, lo, shift):
        return ((int(hi) << 32 | int(lo)) >> (8 * (shift & 3))) & 0xFFFFFFFF

    def perm(s0, s1, sel):
        src = [(s1 >> (8 * i)) & 0xFF for i in range(4)] + [(s0 >> (8 * i)) & 0xFF for i in range(4)]
        out = 0
        for k in range(4):
            code = (sel >> (8 * k)) & 0xFF
            out |= (0 if code == 0x0C else src[code]) << (8 * k)
        return out

    for _ in range(200):
        raw = rng.integers(0, 256, 16, dtype=np.uint8)
        for shift in range(4):
            d = [int.from_bytes(raw[4 * i:4 * i + 4].tobytes(), "little") for i in range(3)]
            x, y = alignbyte(d[1], d[0], shift), alignbyte(d[2], d[1], shift)
            taps = raw[shift:shift + 6]
            for k, sel in enumerate((0x0C030C00, 0x0C040C01, 0x0C050C02)):
                pair = perm(y, x, sel)
                assert pair & 0xFFFF == int(taps[k]) and pair >> 16 == int(taps[3 + k])
