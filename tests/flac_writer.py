"""A small FLAC ENCODER for the tests of the library's FLAC decoder (no FLAC file, encoder or libsndfile exists in the image).
Written from the published format specification; it can emit every construct the decoder handles, chosen per frame:
CONSTANT / VERBATIM / FIXED (order 0-4) / LPC subframes, partitioned Rice residuals with 4- or 5-bit parameters and
escape partitions, wasted bits, independent / left-side / right-side / mid-side stereo, odd last block."""
import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def write(self, v, n):
        v = int(v) & ((1 << n) - 1) if n else 0
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def unary(self, q):
        self.bits += [0] * int(q) + [1]

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def tobytes(self):
        self.align()
        b = np.packbits(np.array(self.bits, dtype=np.uint8))
        return b.tobytes()


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    if v < 0x80:
        return bytes([v])
    out, n = [], 0
    while v >= (0x40 >> n) and n < 5:
        out.append(0x80 | (v & 0x3F))
        v >>= 6
        n += 1
    lead = (0xFF << (7 - n - 0)) & 0xFF
    lead = ((0xFF00 >> (n + 1)) & 0xFF) | v
    return bytes([lead] + out[::-1])


def zigzag(v):
    return (v << 1) ^ (v >> 63) if v < 0 else v << 1


def write_residual(bw, res, blocksize, order, method=0, part_order=0, escape_part=None):
    bw.write(method, 2)
    bw.write(part_order, 4)
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    idx = 0
    for part in range(1 << part_order):
        cnt = (blocksize >> part_order) - (order if part == 0 else 0)
        if part_order == 0:
            cnt = blocksize - order
        seg = [int(v) for v in res[idx:idx + cnt]]
        idx += cnt
        if escape_part == part:
            raw = max([abs(v) for v in seg] + [1]).bit_length() + 1
            bw.write(esc, pbits)
            bw.write(raw, 5)
            for v in seg:
                bw.write(v, raw)
            continue
        mean = max(1.0, float(np.mean([abs(v) for v in seg])) if seg else 1.0)
        k = min(esc - 1, max(0, int(np.log2(mean))))
        bw.write(k, pbits)
        for v in seg:
            u = (v << 1) if v >= 0 else ((-v) << 1) - 1
            bw.unary(u >> k)
            if k:
                bw.write(u & ((1 << k) - 1), k)
    assert idx == blocksize - order


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def write_subframe(bw, x, bps, kind, wasted=0, **kw):
    x = [int(v) for v in x]
    n = len(x)
    bw.write(0, 1)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
        bps -= wasted
    def tail():
        if wasted:
            bw.write(1, 1)
            bw.unary(wasted - 1)
        else:
            bw.write(0, 1)
    if kind == "constant":
        bw.write(0, 6); tail()
        bw.write(x[0], bps)
    elif kind == "verbatim":
        bw.write(1, 6); tail()
        for v in x:
            bw.write(v, bps)
    elif kind.startswith("fixed"):
        order = int(kind[5:])
        bw.write(8 + order, 6); tail()
        for v in x[:order]:
            bw.write(v, bps)
        c = FIXED[order]
        res = [x[i] - sum(c[j] * x[i - 1 - j] for j in range(order)) for i in range(order, n)]
        write_residual(bw, res, n, order, **kw)
    elif kind == "lpc":
        coefs, shift, prec = kw.pop("coefs"), kw.pop("shift"), kw.pop("prec")
        order = len(coefs)
        bw.write(31 + order, 6); tail()
        for v in x[:order]:
            bw.write(v, bps)
        bw.write(prec - 1, 4)
        bw.write(shift, 5)
        for c in coefs:
            bw.write(c, prec)
        res = [x[i] - (sum(coefs[j] * x[i - 1 - j] for j in range(order)) >> shift) for i in range(order, n)]
        write_residual(bw, res, n, order, **kw)
    else:
        raise ValueError(kind)


def encode(samples, fs, bps, frames):
    """samples int [T, ch]; frames = list of (blocksize, channel_mode, [per-channel (kind, kwargs)]) covering T in order.
    channel_mode: 'indep' | 'left_side' | 'right_side' | 'mid_side'."""
    samples = np.asarray(samples, dtype=np.int64)
    T, ch = samples.shape
    blocks = [f[0] for f in frames]
    assert sum(blocks) == T
    si = BitWriter()
    si.write(min(blocks[:-1] or blocks), 16); si.write(max(blocks), 16)
    si.write(0, 24); si.write(0, 24)
    si.write(fs, 20); si.write(ch - 1, 3); si.write(bps - 1, 5); si.write(T, 36)
    for _ in range(16):
        si.write(0, 8)
    body = si.tobytes()
    out = bytearray(b"fLaC" + bytes([0x80 | 0, 0, 0, len(body)]) + body)
    pos = 0
    for fno, (bs, mode, subs) in enumerate(frames):
        blk = samples[pos:pos + bs]
        pos += bs
        hdr = BitWriter()
        hdr.write(0b11111111111110, 14); hdr.write(0, 1); hdr.write(0, 1)
        codes = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12}
        bs_code = codes.get(bs, 6 if bs <= 256 else 7)
        hdr.write(bs_code, 4); hdr.write(0, 4)
        ch_code = {"indep": ch - 1, "left_side": 8, "right_side": 9, "mid_side": 10}[mode]
        hdr.write(ch_code, 4); hdr.write(0, 3); hdr.write(0, 1)
        hb = bytearray(hdr.tobytes()) + utf8_number(fno)
        if bs_code == 6:
            hb += bytes([bs - 1])
        elif bs_code == 7:
            hb += bytes([(bs - 1) >> 8, (bs - 1) & 0xFF])
        hb += bytes([crc8(hb)])
        bw = BitWriter()
        chans = [blk[:, c] for c in range(ch)]
        widths = [bps] * ch
        if mode == "left_side":
            chans, widths = [blk[:, 0], blk[:, 0] - blk[:, 1]], [bps, bps + 1]
        elif mode == "right_side":
            chans, widths = [blk[:, 0] - blk[:, 1], blk[:, 1]], [bps + 1, bps]
        elif mode == "mid_side":
            chans, widths = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]], [bps, bps + 1]
        for c in range(ch):
            kind, kw = subs[c]
            write_subframe(bw, chans[c], widths[c], kind, **dict(kw))
        fb = hb + bw.tobytes()
        c16 = crc16(fb)
        out += fb + bytes([c16 >> 8, c16 & 0xFF])
    return bytes(out)
