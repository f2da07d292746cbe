// FLAC stream decoder (HOST code; compiled into liburse_hip.so next to the kernels so that the data path needs no libsndfile).
// The reference reads its corpora with soundfile.read (baseline_code/dataset.py:318-322, simulation/simulate_data_from_param.py:
// 347-349); URGENT speech sources are largely FLAC.  Covers the format as published (xiph.org FLAC format specification): STREAMINFO,
// fixed and variable block sizes, CONSTANT / VERBATIM / FIXED (order 0-4) / LPC (order 1-32) subframes, partitioned Rice residuals
// (4- and 5-bit parameters, escape codes), wasted bits, independent / left-side / right-side / mid-side stereo, 4-32 bit samples,
// up to 8 channels.  CRCs are not verified.  Pointers are HOST pointers.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "urse_common.h"

namespace urse {

struct BitReader {
  const uint8_t* p;
  size_t nbytes, pos;      // pos in bits
  bool fail;
  inline uint32_t bit() {
    if ((pos >> 3) >= nbytes) { fail = true; return 0; }
    const uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
    ++pos;
    return b;
  }
  inline uint64_t read(int n) {            // n <= 57
    uint64_t v = 0;
    while (n > 0) {
      if ((pos >> 3) >= nbytes) { fail = true; return 0; }
      const int avail = 8 - (int)(pos & 7);
      const int take = n < avail ? n : avail;
      const uint32_t byte = p[pos >> 3];
      v = (v << take) | ((byte >> (avail - take)) & ((1u << take) - 1u));
      pos += take;
      n -= take;
    }
    return v;
  }
  inline int64_t read_signed(int n) {
    if (n == 0) return 0;
    const uint64_t v = read(n);
    const uint64_t sign = 1ull << (n - 1);
    return (int64_t)((v ^ sign) - sign);
  }
  inline uint32_t unary() {                // zeros before the next one bit
    uint32_t q = 0;
    while (!fail && bit() == 0) ++q;
    return q;
  }
  inline void align() { pos = (pos + 7) & ~(size_t)7; }
};

struct FlacInfo { int fs, channels, bps; int64_t total; int min_block, max_block; size_t first_frame; };

static int flac_header(const uint8_t* d, size_t n, FlacInfo* info) {
  if (n < 42 || memcmp(d, "fLaC", 4) != 0) { set_error("flac: not a FLAC stream"); return URSE_ERR_INVALID_ARG; }
  size_t pos = 4;
  bool have = false;
  while (pos + 4 <= n) {
    const bool last = d[pos] & 0x80;
    const int type = d[pos] & 0x7f;
    const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
    pos += 4;
    if (pos + len > n) { set_error("flac: truncated metadata"); return URSE_ERR_INVALID_ARG; }
    if (type == 0 && len >= 34) {
      BitReader br{d + pos, len, 0, false};
      info->min_block = (int)br.read(16); info->max_block = (int)br.read(16);
      br.read(24); br.read(24);
      info->fs = (int)br.read(20); info->channels = (int)br.read(3) + 1; info->bps = (int)br.read(5) + 1;
      info->total = (int64_t)br.read(36);
      have = true;
    }
    pos += len;
    if (last) break;
  }
  if (!have) { set_error("flac: no STREAMINFO block"); return URSE_ERR_INVALID_ARG; }
  info->first_frame = pos;
  return URSE_OK;
}

static bool rice_residual(BitReader& br, int blocksize, int order, int32_t* res) {
  const int method = (int)br.read(2);
  if (method > 1) return false;
  const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
  const int po = (int)br.read(4);
  const int parts = 1 << po;
  int idx = 0;
  for (int part = 0; part < parts; ++part) {
    int cnt = (blocksize >> po) - (part == 0 ? order : 0);
    if (po == 0) cnt = blocksize - order;
    if (cnt < 0) return false;
    const int k = (int)br.read(pbits);
    if (k == esc) {
      const int raw = (int)br.read(5);
      for (int i = 0; i < cnt; ++i) res[idx++] = (int32_t)br.read_signed(raw);
    } else {
      for (int i = 0; i < cnt; ++i) {
        const uint32_t q = br.unary();
        const uint32_t u = (q << k) | (k ? (uint32_t)br.read(k) : 0u);
        res[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
    if (br.fail) return false;
  }
  return idx == blocksize - order;
}

static bool subframe(BitReader& br, int blocksize, int bps, int64_t* out, std::vector<int32_t>& res) {
  if (br.bit() != 0) return false;
  const int type = (int)br.read(6);
  int wasted = 0;
  if (br.bit()) wasted = (int)br.unary() + 1;
  bps -= wasted;
  if (bps <= 0) return false;
  if (type == 0) {
    const int64_t v = br.read_signed(bps);
    for (int i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) out[i] = br.read_signed(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) out[i] = br.read_signed(bps);
    if (!rice_residual(br, blocksize, order, res.data())) return false;
    for (int i = order; i < blocksize; ++i) {
      int64_t pred = 0;
      switch (order) {
        case 1: pred = out[i - 1]; break;
        case 2: pred = 2 * out[i - 1] - out[i - 2]; break;
        case 3: pred = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
        case 4: pred = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
        default: break;
      }
      out[i] = pred + res[i - order];
    }
  } else if (type >= 32) {
    const int order = type - 31;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) out[i] = br.read_signed(bps);
    const int prec = (int)br.read(4) + 1;
    if (prec == 16) return false;
    const int shift = (int)br.read_signed(5);
    if (shift < 0) return false;
    int32_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = (int32_t)br.read_signed(prec);
    if (!rice_residual(br, blocksize, order, res.data())) return false;
    for (int i = order; i < blocksize; ++i) {
      int64_t acc = 0;
      for (int j = 0; j < order; ++j) acc += (int64_t)coef[j] * out[i - 1 - j];
      out[i] = (acc >> shift) + res[i - order];
    }
  } else {
    return false;
  }
  if (wasted) for (int i = 0; i < blocksize; ++i) out[i] *= (1LL << wasted);
  return !br.fail;
}

}  // namespace urse

using namespace urse;

// info (host int64 [6]) = {sample rate, channels, bits per sample, total samples (0 = unknown), min block, max block}
extern "C" int urse_flac_info(const void* data, int64_t nbytes, int64_t* info) {
  URSE_CHECK_ARG(data && info && nbytes > 0, "urse_flac_info: bad argument");
  FlacInfo fi;
  int rc = flac_header((const uint8_t*)data, (size_t)nbytes, &fi);
  if (rc) return rc;
  info[0] = fi.fs; info[1] = fi.channels; info[2] = fi.bps; info[3] = fi.total; info[4] = fi.min_block; info[5] = fi.max_block;
  return URSE_OK;
}

// out (host int32 [capacity_frames, channels], interleaved); *decoded = sample frames written
extern "C" int urse_flac_decode(const void* data, int64_t nbytes, int32_t* out, int64_t capacity_frames, int64_t* decoded) {
  URSE_CHECK_ARG(data && out && decoded && nbytes > 0 && capacity_frames > 0, "urse_flac_decode: bad argument");
  const uint8_t* d = (const uint8_t*)data;
  FlacInfo fi;
  int rc = flac_header(d, (size_t)nbytes, &fi);
  if (rc) return rc;
  const int ch = fi.channels;
  std::vector<int64_t> buf((size_t)ch * 65536);
  std::vector<int32_t> res(65536);
  size_t pos = fi.first_frame;
  int64_t written = 0;
  while (pos + 6 <= (size_t)nbytes && written < capacity_frames) {
    if (d[pos] != 0xFF || (d[pos + 1] & 0xFE) != 0xF8) { ++pos; continue; }        // resynchronise
    BitReader br{d, (size_t)nbytes, (pos + 2) * 8, false};
    const int bs_code = (int)br.read(4), sr_code = (int)br.read(4), ch_code = (int)br.read(4), ss_code = (int)br.read(3);
    br.read(1);
    // UTF-8-style coded frame / sample number
    int lead = 0;
    uint32_t b0 = (uint32_t)br.read(8);
    while (b0 & 0x80) { ++lead; b0 <<= 1; b0 &= 0xFF; }
    for (int i = 1; i < lead; ++i) br.read(8);
    int blocksize = 0;
    if (bs_code == 1) blocksize = 192;
    else if (bs_code >= 2 && bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.read(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.read(16) + 1;
    else if (bs_code >= 8) blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.read(8); else if (sr_code == 13 || sr_code == 14) br.read(16);
    br.read(8);                                                                     // CRC-8
    static const int SS[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    const int bps = ss_code == 0 ? fi.bps : SS[ss_code];
    if (blocksize <= 0 || blocksize > 65535 || bps <= 0 || br.fail || ch_code > 10 || (ch_code < 8 && ch_code + 1 != ch) ||
        (ch_code >= 8 && ch != 2)) { ++pos; continue; }
    bool ok = true;
    for (int c = 0; c < ch && ok; ++c) {
      int sbps = bps;
      if ((ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1)) sbps += 1;     // the side channel
      ok = subframe(br, blocksize, sbps, buf.data() + (size_t)c * 65536, res);
    }
    if (!ok) { set_error("flac: corrupt frame at byte %zu", pos); return URSE_ERR_RUNTIME; }
    br.align();
    br.read(16);                                                                    // CRC-16
    int64_t* c0 = buf.data();
    int64_t* c1 = buf.data() + 65536;
    if (ch_code == 8) for (int i = 0; i < blocksize; ++i) c1[i] = c0[i] - c1[i];                       // left, side
    else if (ch_code == 9) for (int i = 0; i < blocksize; ++i) c0[i] = c0[i] + c1[i];                  // side, right
    else if (ch_code == 10) for (int i = 0; i < blocksize; ++i) {                                      // mid, side
      const int64_t side = c1[i];
      const int64_t mid = (c0[i] << 1) | (side & 1);
      c0[i] = (mid + side) >> 1;
      c1[i] = (mid - side) >> 1;
    }
    const int64_t n = written + blocksize <= capacity_frames ? blocksize : capacity_frames - written;
    for (int64_t i = 0; i < n; ++i)
      for (int c = 0; c < ch; ++c) out[(written + i) * ch + c] = (int32_t)buf[(size_t)c * 65536 + i];
    written += n;
    pos = br.pos >> 3;
  }
  *decoded = written;
  return URSE_OK;
}
