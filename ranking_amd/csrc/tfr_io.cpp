// libtfr_io.so -- host-side input path of the ranking framework (see include/tfr_io.h):
// TFRecord framing, ExampleListWithContext -> padded dense fp32 tensors, LibSVM loader.
//
// Reference behaviour restated: python/data.py:59-96 (ELWC wire format: `repeated bytes
// examples = 1; bytes context = 2`, each a serialized tf.Example), :133-208 (truncate / pad to
// list_size, sizes, mask), :383-540; examples/tf_ranking_libsvm.py:137-195 (LibSVM loader).
// The reference delegates the byte work to TensorFlow C++ ops (tf.io.parse_example,
// TFRecordDataset); here it is a small protobuf wire reader with no dependencies.
#include "../../include/tfr_io.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ------------------------------------------------------------------ crc32c
struct Crc32cTable {
  uint32_t t[8][256];
  Crc32cTable() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
  }
};
const Crc32cTable& crc_table() { static const Crc32cTable tab; return tab; }

uint32_t crc32c_sliced(const uint8_t* p, size_t n) {
  const Crc32cTable& T = crc_table();
  uint32_t c = 0xffffffffu;
  while (n >= 8) {                                   // slice-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^
        T.t[3][hi & 0xff] ^ T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
    p += 8; n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}
#if defined(__x86_64__)
// The SSE4.2 crc32 instruction computes exactly this polynomial (Castagnoli): 8 bytes per 3-cycle issue, ~5x
// the table walk.  Chosen once at run time; the table version stays as the portable path and the cross-check.
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const uint8_t* p, size_t n) {
  uint64_t c = 0xffffffffu;
  while (n >= 32) {
    uint64_t a, b, d, e;
    memcpy(&a, p, 8); memcpy(&b, p + 8, 8); memcpy(&d, p + 16, 8); memcpy(&e, p + 24, 8);
    c = __builtin_ia32_crc32di(c, a); c = __builtin_ia32_crc32di(c, b);
    c = __builtin_ia32_crc32di(c, d); c = __builtin_ia32_crc32di(c, e);
    p += 32; n -= 32;
  }
  while (n >= 8) { uint64_t a; memcpy(&a, p, 8); c = __builtin_ia32_crc32di(c, a); p += 8; n -= 8; }
  uint32_t c32 = (uint32_t)c;
  while (n--) c32 = __builtin_ia32_crc32qi(c32, *p++);
  return c32 ^ 0xffffffffu;
}
uint32_t crc32c(const uint8_t* p, size_t n) {
  static const bool hw = __builtin_cpu_supports("sse4.2");
  return hw ? crc32c_hw(p, n) : crc32c_sliced(p, n);
}
#else
uint32_t crc32c(const uint8_t* p, size_t n) { return crc32c_sliced(p, n); }
#endif
inline uint32_t mask_crc(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

// ------------------------------------------------------------------ protobuf wire reader
struct Reader {
  const uint8_t* p; const uint8_t* end; bool ok = true;
  Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  bool done() const { return p >= end || !ok; }
  uint64_t varint() {
    uint64_t v = 0; int shift = 0;
    while (p < end && shift < 64) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
      shift += 7;
    }
    ok = false; return 0;
  }
  // length-delimited payload
  bool bytes(const uint8_t*& b, size_t& n) {
    const uint64_t len = varint();
    if (!ok || len > (uint64_t)(end - p)) { ok = false; return false; }
    b = p; n = (size_t)len; p += len; return true;
  }
  void skip(uint32_t wire) {
    switch (wire) {
      case 0: varint(); break;
      case 1: if (end - p < 8) ok = false; else p += 8; break;
      case 2: { const uint8_t* b; size_t n; bytes(b, n); break; }
      case 5: if (end - p < 4) ok = false; else p += 4; break;
      default: ok = false;
    }
  }
};

struct SpecTable {
  std::unordered_map<std::string_view, int> index;       // name -> spec index
  std::vector<int> offset;                                // column offset of each spec
  std::vector<int> width;
  std::vector<float> dflt;
  std::vector<std::string_view> name;
  std::vector<float> default_row;                         // one padded example
  int total = 0;
  SpecTable(const tfr_io_feature_spec* specs, int n) {
    offset.resize(n); width.resize(n); dflt.resize(n); name.resize(n);
    for (int i = 0; i < n; ++i) {
      name[i] = std::string_view(specs[i].name);
      index.emplace(name[i], i);
      offset[i] = total; width[i] = specs[i].width; dflt[i] = specs[i].default_value;
      total += specs[i].width;
    }
    default_row.resize(total > 0 ? total : 1);
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < width[i]; ++k) default_row[offset[i] + k] = dflt[i];
  }
  void fill_defaults(float* row) const { memcpy(row, default_row.data(), (size_t)total * sizeof(float)); }
  // Examples of one file serialize their features in one order: `hint[i]` remembers which spec the i-th map
  // entry of the previous example was, so the usual lookup is one length + memcmp check instead of a string
  // hash (-1 = a feature the spec does not name).  Per-thread state, owned by the caller.
  int lookup(std::string_view key, size_t pos, std::vector<int>& hint) const {
    if (pos < hint.size()) {
      const int h = hint[pos];
      if (h >= 0 && name[h].size() == key.size() && memcmp(name[h].data(), key.data(), key.size()) == 0) return h;
    } else {
      hint.resize(pos + 1, -2);
    }
    const auto it = index.find(key);
    const int s = it == index.end() ? -1 : it->second;
    hint[pos] = s;
    return s;
  }
};

// Feature { oneof kind { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
// Writes exactly `width` values; returns 0, TFR_IO_ESHAPE, TFR_IO_ETYPE or TFR_IO_ECORRUPT.
// `payload_at`: set to the offset of the 4 * width payload bytes inside `b` when one of the two fixed-image fast
// paths decoded the feature (the only cases the example template below can replay), -1 otherwise.
int decode_feature(const uint8_t* b, size_t n, int width, float* out, int* payload_at) {
  // Fast paths for what TensorFlow writes for a float feature of `width` values, all lengths < 128:
  //   packed    12 <4w+2> 0a <4w> <4w bytes>          unpacked (w = 1)   12 05 0d <4 bytes>
  const size_t pw = (size_t)width * 4;
  *payload_at = -1;
  if (pw + 2 < 128 && n == pw + 4 && b[0] == 0x12 && b[1] == pw + 2 && b[2] == 0x0a && b[3] == pw) {
    memcpy(out, b + 4, pw);
    *payload_at = 4;
    return 0;
  }
  if (width == 1 && n == 7 && b[0] == 0x12 && b[1] == 5 && b[2] == 0x0d) {
    memcpy(out, b + 3, 4);
    *payload_at = 3;
    return 0;
  }
  Reader r(b, n);
  int count = 0;
  bool seen = false;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) return TFR_IO_ECORRUPT;
    const uint32_t field = (uint32_t)(tag >> 3), wire = (uint32_t)(tag & 7);
    if (wire != 2) { r.skip(wire); continue; }
    const uint8_t* lb; size_t ln;
    if (!r.bytes(lb, ln)) return TFR_IO_ECORRUPT;
    if (field == 1) {                                     // bytes_list
      if (ln > 0) return TFR_IO_ETYPE;                    // an empty list is "feature absent"
      continue;
    }
    if (field != 2 && field != 3) continue;
    seen = true;
    Reader l(lb, ln);
    while (!l.done()) {
      const uint64_t t2 = l.varint();
      if (!l.ok) return TFR_IO_ECORRUPT;
      const uint32_t f2 = (uint32_t)(t2 >> 3), w2 = (uint32_t)(t2 & 7);
      if (f2 != 1) { l.skip(w2); continue; }
      if (field == 2) {                                   // FloatList.value (packed or not)
        if (w2 == 2) {
          const uint8_t* pb; size_t pn;
          if (!l.bytes(pb, pn) || (pn & 3)) return TFR_IO_ECORRUPT;
          const size_t vals = pn / 4;
          if (count < width) memcpy(out + count, pb, std::min(vals, (size_t)(width - count)) * 4);
          count += (int)vals;
        } else if (w2 == 5) {
          if (l.end - l.p < 4) return TFR_IO_ECORRUPT;
          float v; memcpy(&v, l.p, 4); l.p += 4;
          if (count < width) out[count] = v;
          ++count;
        } else {
          return TFR_IO_ECORRUPT;
        }
      } else {                                            // Int64List.value (packed or not)
        if (w2 == 2) {
          const uint8_t* pb; size_t pn;
          if (!l.bytes(pb, pn)) return TFR_IO_ECORRUPT;
          Reader pv(pb, pn);
          while (!pv.done()) {
            const int64_t v = (int64_t)pv.varint();
            if (!pv.ok) return TFR_IO_ECORRUPT;
            if (count < width) out[count] = (float)v;
            ++count;
          }
        } else if (w2 == 0) {
          const int64_t v = (int64_t)l.varint();
          if (!l.ok) return TFR_IO_ECORRUPT;
          if (count < width) out[count] = (float)v;
          ++count;
        } else {
          return TFR_IO_ECORRUPT;
        }
      }
    }
    if (!l.ok) return TFR_IO_ECORRUPT;
  }
  if (!r.ok) return TFR_IO_ECORRUPT;
  if (!seen || count == 0) return 1;                      // absent (or empty list): keep the default
  return count == width ? 0 : TFR_IO_ESHAPE;
}

#if defined(__x86_64__)
// ((a ^ t) & m) == 0 over `len` bytes, 64 bytes per iteration (chosen at run time like the crc above)
__attribute__((target("avx2"))) bool masked_equal_avx2(const uint8_t* a, const uint8_t* t, const uint8_t* m, size_t len) {
  typedef long long v4 __attribute__((vector_size(32), aligned(1), may_alias));
  v4 acc = {0, 0, 0, 0};
  size_t i = 0;
  for (; i + 64 <= len; i += 64) {
    const v4 a0 = *reinterpret_cast<const v4*>(a + i), a1 = *reinterpret_cast<const v4*>(a + i + 32);
    const v4 t0 = *reinterpret_cast<const v4*>(t + i), t1 = *reinterpret_cast<const v4*>(t + i + 32);
    const v4 m0 = *reinterpret_cast<const v4*>(m + i), m1 = *reinterpret_cast<const v4*>(m + i + 32);
    acc |= ((a0 ^ t0) & m0) | ((a1 ^ t1) & m1);
  }
  uint64_t tail = 0;
  for (; i < len; ++i) tail |= (uint64_t)((a[i] ^ t[i]) & m[i]);
  return (acc[0] | acc[1] | acc[2] | acc[3] | (long long)tail) == 0;
}
#endif

// Example template.  The examples of one file are written by one writer: same features, same order, same lengths --
// only the value bytes differ.  After a generic parse in which every feature the spec names came through one of the
// fixed-image fast paths of decode_feature, the example's bytes are kept together with a mask that is zero on every
// byte the generic parser never interprets (the float payloads of the named features, the whole value of the features
// the spec does not name) and a list of (payload offset -> column) copies.  The next example of the same length whose
// unmasked bytes are identical has, byte for byte, the structure the generic parser would walk -- its control flow
// depends on nothing else -- so the parse is: defaults, then the copies, in the same order (a repeated key still lets
// the later entry win).  Anything else (a missing feature, an int64 list, another length) falls back to the generic
// parse, which re-arms the template.  Per worker thread; `streak` stops the re-arming on data that never repeats.
struct ExampleTemplate {
  struct Copy { uint32_t src, dst, len; };
  std::vector<uint8_t> bytes, mask;                       // padded to a multiple of 8
  std::vector<Copy> copies;
  size_t n = 0;
  bool valid = false;
  int streak = 0;                                         // consecutive generic parses that matched no template
  uint64_t replayed = 0, walked = 0;                      // examples parsed by replay / by the generic walk (this thread)
  // scratch of the generic parse that may arm the template
  std::vector<Copy> rec_copies;
  std::vector<std::pair<uint32_t, uint32_t>> rec_skips;  // (offset, length) of never-interpreted value bytes
  bool matches(const uint8_t* b, size_t len) const {
    if (!valid || len != n) return false;
    const uint8_t* t = bytes.data(); const uint8_t* m = mask.data();
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return masked_equal_avx2(b, t, m, len);
#endif
    uint64_t acc = 0;
    size_t i = 0;
    for (; i + 32 <= len; i += 32) {
      uint64_t a0, a1, a2, a3, t0, t1, t2, t3, m0, m1, m2, m3;
      memcpy(&a0, b + i, 8); memcpy(&a1, b + i + 8, 8); memcpy(&a2, b + i + 16, 8); memcpy(&a3, b + i + 24, 8);
      memcpy(&t0, t + i, 8); memcpy(&t1, t + i + 8, 8); memcpy(&t2, t + i + 16, 8); memcpy(&t3, t + i + 24, 8);
      memcpy(&m0, m + i, 8); memcpy(&m1, m + i + 8, 8); memcpy(&m2, m + i + 16, 8); memcpy(&m3, m + i + 24, 8);
      acc |= ((a0 ^ t0) & m0) | ((a1 ^ t1) & m1) | ((a2 ^ t2) & m2) | ((a3 ^ t3) & m3);
    }
    for (; i < len; ++i) acc |= (uint64_t)((b[i] ^ t[i]) & m[i]);
    return acc == 0;
  }
  void arm(const uint8_t* b, size_t len) {
    n = len;
    bytes.assign(b, b + len); bytes.resize((len + 7) / 8 * 8 + 32, 0);
    mask.assign(bytes.size(), 0);
    memset(mask.data(), 0xff, len);
    for (const Copy& c : rec_copies) memset(mask.data() + c.src, 0, c.len);
    for (const auto& sk : rec_skips) memset(mask.data() + sk.first, 0, sk.second);
    copies = rec_copies;
    valid = true;
  }
};

std::atomic<uint64_t> g_examples_replayed{0}, g_examples_walked{0};   // process totals (tfr_io_parse_counters)

bool template_enabled() {
  static const bool on = [] { const char* e = getenv("TFR_IO_TEMPLATE"); return !(e && e[0] == '0'); }();
  return on;
}

// tf.Example { Features features = 1; }  Features { map<string, Feature> feature = 1; }
int decode_example(const uint8_t* b, size_t n, const SpecTable& specs, float* row, std::vector<int>& hint,
                   ExampleTemplate& tpl) {
  specs.fill_defaults(row);
  const bool use_tpl = template_enabled() && n < (1u << 30);
  if (use_tpl && tpl.matches(b, n)) {
    for (const ExampleTemplate::Copy& c : tpl.copies) {
      if (c.len == 4) { uint32_t v; memcpy(&v, b + c.src, 4); memcpy(row + c.dst, &v, 4); }   // the scalar feature
      else memcpy(row + c.dst, b + c.src, c.len);
    }
    tpl.streak = 0;
    ++tpl.replayed;
    return 0;
  }
  ++tpl.walked;
  const bool record = use_tpl && tpl.streak < 16;         // data that never repeats: stop paying for the copies
  bool replayable = record;
  if (record) { tpl.rec_copies.clear(); tpl.rec_skips.clear(); }
  size_t pos = 0;                                         // index of the map entry inside this example
  Reader r(b, n);
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) return TFR_IO_ECORRUPT;
    if ((tag >> 3) != 1 || (tag & 7) != 2) { r.skip((uint32_t)(tag & 7)); continue; }
    const uint8_t* fb; size_t fn;
    if (!r.bytes(fb, fn)) return TFR_IO_ECORRUPT;
    Reader f(fb, fn);                                     // Features
    while (!f.done()) {
      const uint64_t t2 = f.varint();
      if (!f.ok) return TFR_IO_ECORRUPT;
      if ((t2 >> 3) != 1 || (t2 & 7) != 2) { f.skip((uint32_t)(t2 & 7)); continue; }
      const uint8_t* eb; size_t en;
      if (!f.bytes(eb, en)) return TFR_IO_ECORRUPT;
      Reader e(eb, en);                                   // map entry { key = 1; value = 2; }
      std::string_view key; const uint8_t* vb = nullptr; size_t vn = 0;
      // the usual image: 0a <klen> key 12 <vlen> value, both lengths one byte
      if (en >= 4 && eb[0] == 0x0a && eb[1] < 128) {
        const size_t kl = eb[1];
        if (kl + 4 <= en && eb[2 + kl] == 0x12 && eb[3 + kl] < 128 && kl + 4 + eb[3 + kl] == en) {
          key = std::string_view(reinterpret_cast<const char*>(eb + 2), kl);
          vb = eb + 4 + kl; vn = eb[3 + kl];
          e.p = e.end;                                    // consumed
        }
      }
      while (!e.done()) {
        const uint64_t t3 = e.varint();
        if (!e.ok) return TFR_IO_ECORRUPT;
        const uint32_t f3 = (uint32_t)(t3 >> 3), w3 = (uint32_t)(t3 & 7);
        if (w3 != 2) { e.skip(w3); continue; }
        const uint8_t* xb; size_t xn;
        if (!e.bytes(xb, xn)) return TFR_IO_ECORRUPT;
        if (f3 == 1) key = std::string_view(reinterpret_cast<const char*>(xb), xn);
        else if (f3 == 2) { vb = xb; vn = xn; }
      }
      if (!e.ok) return TFR_IO_ECORRUPT;
      const int s = specs.lookup(key, pos++, hint);
      if (s < 0 || vb == nullptr) {
        if (replayable && vb != nullptr && vn > 0)        // a feature the spec does not name: its value is never read
          tpl.rec_skips.emplace_back((uint32_t)(vb - b), (uint32_t)vn);
        continue;
      }
      int payload_at = -1;
      const int rc = decode_feature(vb, vn, specs.width[s], row + specs.offset[s], &payload_at);
      if (rc < 0) return rc;
      if (rc == 1)                                        // present but empty: default
        for (int k = 0; k < specs.width[s]; ++k) row[specs.offset[s] + k] = specs.dflt[s];
      if (replayable) {
        if (payload_at < 0) replayable = false;
        else tpl.rec_copies.push_back({(uint32_t)(vb - b) + (uint32_t)payload_at, (uint32_t)specs.offset[s],
                                       (uint32_t)specs.width[s] * 4u});
      }
    }
    if (!f.ok) return TFR_IO_ECORRUPT;
  }
  if (!r.ok) return TFR_IO_ECORRUPT;
  if (use_tpl) {
    ++tpl.streak;
    if (replayable) tpl.arm(b, n); else tpl.valid = false;
  }
  return 0;
}

// ExampleListWithContext { repeated bytes examples = 1; bytes context = 2; }   (data.py:59-77)
struct Hints { std::vector<int> example, context; ExampleTemplate example_tpl, context_tpl; };   // per worker thread

int decode_elwc(const uint8_t* b, size_t n, int list_size, const SpecTable& ex, const SpecTable* ctx,
                float* example_rows, float* context_row, int32_t* size_out, uint8_t* mask_row, Hints& hints) {
  Reader r(b, n);
  int count = 0;
  bool ctx_seen = false;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) return TFR_IO_ECORRUPT;
    const uint32_t field = (uint32_t)(tag >> 3), wire = (uint32_t)(tag & 7);
    if (wire != 2 || (field != 1 && field != 2)) { r.skip(wire); continue; }
    const uint8_t* pb; size_t pn;
    if (!r.bytes(pb, pn)) return TFR_IO_ECORRUPT;
    if (field == 1) {
      if (count < list_size) {                            // truncation keeps the first list_size (:170-172)
        const int rc = decode_example(pb, pn, ex, example_rows + (size_t)count * ex.total, hints.example, hints.example_tpl);
        if (rc < 0) return rc;
      }
      ++count;
    } else if (ctx && context_row) {
      const int rc = decode_example(pb, pn, *ctx, context_row, hints.context, hints.context_tpl);
      if (rc < 0) return rc;
      ctx_seen = true;
    }
  }
  if (!r.ok) return TFR_IO_ECORRUPT;
  for (int i = std::min(count, list_size); i < list_size; ++i)   // padding = empty Example = defaults (:174-178)
    ex.fill_defaults(example_rows + (size_t)i * ex.total);
  if (ctx && context_row && !ctx_seen) ctx->fill_defaults(context_row);
  if (size_out) *size_out = count;
  if (mask_row)
    for (int i = 0; i < list_size; ++i) mask_row[i] = i < count ? 1 : 0;
  return 0;
}

int count_examples(const uint8_t* b, size_t n) {
  Reader r(b, n);
  int count = 0;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) return TFR_IO_ECORRUPT;
    const uint32_t wire = (uint32_t)(tag & 7);
    if ((tag >> 3) == 1 && wire == 2) ++count;
    r.skip(wire);
  }
  return r.ok ? count : TFR_IO_ECORRUPT;
}

// ---- ExampleInExample (data.py:136-151, 211-380): a tf.Example with two bytes features, `serialized_context` (exactly
// one serialized tf.Example: FixedLenFeature([1], string) in the reference, a record without it is an error there and
// TFR_IO_EMISSING here) and `serialized_examples` (one serialized tf.Example per item; "" is an example of defaults).
// Walks the outer map; `on_context` / `on_example` receive the byte ranges.  Returns 0 or a negative TFR_IO_E*.
template <typename FC, typename FE>
int walk_eie(const uint8_t* b, size_t n, FC&& on_context, FE&& on_example) {
  bool ctx_seen = false;
  Reader r(b, n);
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) return TFR_IO_ECORRUPT;
    if ((tag >> 3) != 1 || (tag & 7) != 2) { r.skip((uint32_t)(tag & 7)); continue; }
    const uint8_t* fb; size_t fn;
    if (!r.bytes(fb, fn)) return TFR_IO_ECORRUPT;
    Reader f(fb, fn);                                     // Features
    while (!f.done()) {
      const uint64_t t2 = f.varint();
      if (!f.ok) return TFR_IO_ECORRUPT;
      if ((t2 >> 3) != 1 || (t2 & 7) != 2) { f.skip((uint32_t)(t2 & 7)); continue; }
      const uint8_t* eb; size_t en;
      if (!f.bytes(eb, en)) return TFR_IO_ECORRUPT;
      Reader e(eb, en);                                   // map entry { key = 1; value = 2; }
      std::string_view key; const uint8_t* vb = nullptr; size_t vn = 0;
      while (!e.done()) {
        const uint64_t t3 = e.varint();
        if (!e.ok) return TFR_IO_ECORRUPT;
        const uint32_t f3 = (uint32_t)(t3 >> 3), w3 = (uint32_t)(t3 & 7);
        if (w3 != 2) { e.skip(w3); continue; }
        const uint8_t* xb; size_t xn;
        if (!e.bytes(xb, xn)) return TFR_IO_ECORRUPT;
        if (f3 == 1) key = std::string_view(reinterpret_cast<const char*>(xb), xn);
        else if (f3 == 2) { vb = xb; vn = xn; }
      }
      if (!e.ok) return TFR_IO_ECORRUPT;
      const bool is_ctx = key == "serialized_context", is_ex = key == "serialized_examples";
      if ((!is_ctx && !is_ex) || vb == nullptr) continue;
      Reader v(vb, vn);                                   // Feature: bytes_list = 1
      int n_ctx = 0;
      while (!v.done()) {
        const uint64_t t4 = v.varint();
        if (!v.ok) return TFR_IO_ECORRUPT;
        const uint32_t f4 = (uint32_t)(t4 >> 3), w4 = (uint32_t)(t4 & 7);
        if (w4 != 2) { v.skip(w4); continue; }
        const uint8_t* lb; size_t ln;
        if (!v.bytes(lb, ln)) return TFR_IO_ECORRUPT;
        if (f4 != 1) { if (ln > 0) return TFR_IO_ETYPE; continue; }       // a float / int64 list under these keys
        Reader l(lb, ln);                                 // BytesList { repeated bytes value = 1; }
        while (!l.done()) {
          const uint64_t t5 = l.varint();
          if (!l.ok) return TFR_IO_ECORRUPT;
          if ((t5 >> 3) != 1 || (t5 & 7) != 2) { l.skip((uint32_t)(t5 & 7)); continue; }
          const uint8_t* sb; size_t sn;
          if (!l.bytes(sb, sn)) return TFR_IO_ECORRUPT;
          int rc = 0;
          if (is_ctx) { if (++n_ctx > 1) return TFR_IO_ESHAPE; ctx_seen = true; rc = on_context(sb, sn); }
          else rc = on_example(sb, sn);
          if (rc < 0) return rc;
        }
        if (!l.ok) return TFR_IO_ECORRUPT;
      }
      if (!v.ok) return TFR_IO_ECORRUPT;
    }
    if (!f.ok) return TFR_IO_ECORRUPT;
  }
  if (!r.ok) return TFR_IO_ECORRUPT;
  return ctx_seen ? 0 : TFR_IO_EMISSING;
}

int decode_eie(const uint8_t* b, size_t n, int list_size, const SpecTable& ex, const SpecTable* ctx,
               float* example_rows, float* context_row, int32_t* size_out, uint8_t* mask_row, Hints& hints) {
  int count = 0;
  if (ctx && context_row) ctx->fill_defaults(context_row);
  const int rc = walk_eie(
      b, n,
      [&](const uint8_t* sb, size_t sn) {
        return (ctx && context_row) ? decode_example(sb, sn, *ctx, context_row, hints.context, hints.context_tpl) : 0;
      },
      [&](const uint8_t* sb, size_t sn) {
        int r2 = 0;
        if (count < list_size)
          r2 = decode_example(sb, sn, ex, example_rows + (size_t)count * ex.total, hints.example, hints.example_tpl);
        ++count;
        return r2;
      });
  if (rc < 0) return rc;
  for (int i = std::min(count, list_size); i < list_size; ++i) ex.fill_defaults(example_rows + (size_t)i * ex.total);
  if (size_out) *size_out = count;
  if (mask_row)
    for (int i = 0; i < list_size; ++i) mask_row[i] = i < count ? 1 : 0;
  return 0;
}

int count_eie(const uint8_t* b, size_t n) {
  int count = 0;
  const int rc = walk_eie(b, n, [](const uint8_t*, size_t) { return 0; }, [&](const uint8_t*, size_t) { ++count; return 0; });
  return rc < 0 ? rc : count;
}

// ---- tf.SequenceExample (data.py:572-710): { Features context = 1; FeatureLists feature_lists = 2; },
// FeatureLists { map<string, FeatureList> feature_list = 1; }, FeatureList { repeated Feature feature = 1; }.
// Every named example feature is a FixedLenSequenceFeature(allow_missing=True) there: a missing feature_list has no
// frames; frame t of feature k is example t's value of k and must carry exactly `width` values (an empty frame is an
// error, data_test.py:793-819) -- also in the frames the truncation drops; positions past a feature's own frames take
// its default; sizes = the longest named feature list.  `frames_out` (nullable): only count (the list-size query).
int decode_seq(const uint8_t* b, size_t n, int list_size, const SpecTable& ex, const SpecTable* ctx,
               float* example_rows, float* context_row, int32_t* size_out, uint8_t* mask_row, Hints& hints,
               bool count_only) {
  int longest = 0;
  if (!count_only) {
    for (int i = 0; i < list_size; ++i) ex.fill_defaults(example_rows + (size_t)i * ex.total);
    if (ctx && context_row) ctx->fill_defaults(context_row);
  }
  std::vector<float> scratch;
  std::vector<int> frames_of(ex.width.size(), -1);        // a repeated map key: the later entry wins (protobuf maps)
  Reader r(b, n);
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) return TFR_IO_ECORRUPT;
    const uint32_t field = (uint32_t)(tag >> 3), wire = (uint32_t)(tag & 7);
    if (wire != 2 || (field != 1 && field != 2)) { r.skip(wire); continue; }
    const uint8_t* pb; size_t pn;
    if (!r.bytes(pb, pn)) return TFR_IO_ECORRUPT;
    if (field == 1) {                                     // context: a Features message = the payload of Example.features
      if (count_only || !ctx || !context_row) continue;
      // decode_example expects Example { features = 1 }: walk the Features payload with the same entry decoder by
      // re-framing is a copy; instead parse the map entries here (contexts are a handful of values per list)
      Reader f(pb, pn);
      size_t pos = 0;
      while (!f.done()) {
        const uint64_t t2 = f.varint();
        if (!f.ok) return TFR_IO_ECORRUPT;
        if ((t2 >> 3) != 1 || (t2 & 7) != 2) { f.skip((uint32_t)(t2 & 7)); continue; }
        const uint8_t* eb; size_t en;
        if (!f.bytes(eb, en)) return TFR_IO_ECORRUPT;
        Reader e(eb, en);
        std::string_view key; const uint8_t* vb = nullptr; size_t vn = 0;
        while (!e.done()) {
          const uint64_t t3 = e.varint();
          if (!e.ok) return TFR_IO_ECORRUPT;
          const uint32_t f3 = (uint32_t)(t3 >> 3), w3 = (uint32_t)(t3 & 7);
          if (w3 != 2) { e.skip(w3); continue; }
          const uint8_t* xb; size_t xn;
          if (!e.bytes(xb, xn)) return TFR_IO_ECORRUPT;
          if (f3 == 1) key = std::string_view(reinterpret_cast<const char*>(xb), xn);
          else if (f3 == 2) { vb = xb; vn = xn; }
        }
        if (!e.ok) return TFR_IO_ECORRUPT;
        const int s = ctx->lookup(key, pos++, hints.context);
        if (s < 0 || vb == nullptr) continue;
        int payload_at = -1;
        const int rc = decode_feature(vb, vn, ctx->width[s], context_row + ctx->offset[s], &payload_at);
        if (rc < 0) return rc;
        if (rc == 1)
          for (int k = 0; k < ctx->width[s]; ++k) context_row[ctx->offset[s] + k] = ctx->dflt[s];
      }
      if (!f.ok) return TFR_IO_ECORRUPT;
      continue;
    }
    Reader fl(pb, pn);                                    // FeatureLists
    size_t pos = 0;
    while (!fl.done()) {
      const uint64_t t2 = fl.varint();
      if (!fl.ok) return TFR_IO_ECORRUPT;
      if ((t2 >> 3) != 1 || (t2 & 7) != 2) { fl.skip((uint32_t)(t2 & 7)); continue; }
      const uint8_t* eb; size_t en;
      if (!fl.bytes(eb, en)) return TFR_IO_ECORRUPT;
      Reader e(eb, en);                                   // map entry { key = 1; FeatureList value = 2; }
      std::string_view key; const uint8_t* vb = nullptr; size_t vn = 0;
      while (!e.done()) {
        const uint64_t t3 = e.varint();
        if (!e.ok) return TFR_IO_ECORRUPT;
        const uint32_t f3 = (uint32_t)(t3 >> 3), w3 = (uint32_t)(t3 & 7);
        if (w3 != 2) { e.skip(w3); continue; }
        const uint8_t* xb; size_t xn;
        if (!e.bytes(xb, xn)) return TFR_IO_ECORRUPT;
        if (f3 == 1) key = std::string_view(reinterpret_cast<const char*>(xb), xn);
        else if (f3 == 2) { vb = xb; vn = xn; }
      }
      if (!e.ok) return TFR_IO_ECORRUPT;
      const int s = ex.lookup(key, pos++, hints.example);
      if (s < 0) continue;
      const int w = ex.width[s];
      if ((int)scratch.size() < w) scratch.resize(w);
      if (frames_of[s] >= 0 && !count_only)
        for (int i = 0; i < list_size; ++i)
          for (int k = 0; k < w; ++k) example_rows[(size_t)i * ex.total + ex.offset[s] + k] = ex.dflt[s];
      int frame = 0;
      Reader v(vb, vn);                                   // FeatureList { repeated Feature feature = 1; }
      while (vb != nullptr && !v.done()) {
        const uint64_t t4 = v.varint();
        if (!v.ok) return TFR_IO_ECORRUPT;
        if ((t4 >> 3) != 1 || (t4 & 7) != 2) { v.skip((uint32_t)(t4 & 7)); continue; }
        const uint8_t* xb; size_t xn;
        if (!v.bytes(xb, xn)) return TFR_IO_ECORRUPT;
        if (!count_only) {
          float* dst = frame < list_size ? example_rows + (size_t)frame * ex.total + ex.offset[s] : scratch.data();
          int payload_at = -1;
          const int rc = decode_feature(xb, xn, w, dst, &payload_at);
          if (rc < 0) return rc;
          if (rc == 1) return TFR_IO_ESHAPE;              // an empty frame: "values size: 0 but output shape: [w]"
        }
        ++frame;
      }
      if (vb != nullptr && !v.ok) return TFR_IO_ECORRUPT;
      frames_of[s] = frame;
    }
    if (!fl.ok) return TFR_IO_ECORRUPT;
  }
  if (!r.ok) return TFR_IO_ECORRUPT;
  for (int c : frames_of) longest = std::max(longest, c);
  if (size_out) *size_out = longest;
  if (mask_row && !count_only)
    for (int i = 0; i < list_size; ++i) mask_row[i] = i < longest ? 1 : 0;
  return count_only ? longest : 0;
}

}  // namespace

extern "C" int tfr_io_abi_version(void) { return TFR_IO_ABI_VERSION; }

extern "C" uint32_t tfr_io_crc32c(const uint8_t* data, size_t n) { return crc32c(data, n); }
extern "C" uint32_t tfr_io_crc32c_portable(const uint8_t* data, size_t n) { return crc32c_sliced(data, n); }
extern "C" uint32_t tfr_io_masked_crc32c(const uint8_t* data, size_t n) { return mask_crc(crc32c(data, n)); }

extern "C" int64_t tfr_io_tfrecord_index(const uint8_t* buf, size_t nbytes, int verify_crc, uint64_t* offsets,
                                         uint64_t* lengths, int64_t max_records) {
  if (!buf && nbytes) return TFR_IO_EINVAL;
  size_t pos = 0;
  int64_t n = 0;
  while (pos < nbytes) {
    if (nbytes - pos < 12) return TFR_IO_ECORRUPT;
    uint64_t len; uint32_t lcrc;
    memcpy(&len, buf + pos, 8); memcpy(&lcrc, buf + pos + 8, 4);
    if (verify_crc && mask_crc(crc32c(buf + pos, 8)) != lcrc) return TFR_IO_ECRC;
    if (len > nbytes - pos - 12 || nbytes - pos - 12 - len < 4) return TFR_IO_ECORRUPT;
    const size_t data = pos + 12;
    if (verify_crc) {
      uint32_t dcrc; memcpy(&dcrc, buf + data + len, 4);
      if (mask_crc(crc32c(buf + data, (size_t)len)) != dcrc) return TFR_IO_ECRC;
    }
    if (n < max_records) {
      if (offsets) offsets[n] = data;
      if (lengths) lengths[n] = len;
    }
    ++n;
    pos = data + (size_t)len + 4;
  }
  return n;
}

extern "C" int64_t tfr_io_elwc_max_list_size(const uint8_t* const* records, const uint64_t* lengths, int32_t B) {
  if (B < 0 || (B > 0 && (!records || !lengths))) return TFR_IO_EINVAL;
  int64_t best = 0;
  for (int b = 0; b < B; ++b) {
    const int c = count_examples(records[b], (size_t)lengths[b]);
    if (c < 0) return c;
    best = std::max<int64_t>(best, c);
  }
  return best;
}

extern "C" void tfr_io_parse_counters(uint64_t* replayed, uint64_t* walked) {
  if (replayed) *replayed = g_examples_replayed.load(std::memory_order_relaxed);
  if (walked) *walked = g_examples_walked.load(std::memory_order_relaxed);
}

// fp32 -> bf16, round to nearest even; NaN stays NaN (quiet bit set) -- bit for bit what v_cvt_pk_bf16_f32 and
// torch's .to(bfloat16) produce, so features rounded here equal features rounded by the scorer's input cast.
static inline uint16_t bf16_rne(uint32_t u) {              // branch-free: the loop below vectorises
  const uint32_t rounded = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  const uint32_t quiet = (u >> 16) | 0x0040u;
  return (uint16_t)(((u & 0x7fffffffu) > 0x7f800000u) ? quiet : rounded);
}

// one clone per vector ISA, picked at load time (the parser's other loops are byte walks: nothing to gain there)
// (x86-64 GCC only: target_clones needs ifunc support and `optimize` is a GCC attribute -- any other host compiler /
// architecture takes the plain loop at the translation unit's optimisation level)
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && defined(__gnu_linux__)
__attribute__((target_clones("arch=skylake-avx512", "avx2", "default"), optimize("O3")))
#endif
void f32_to_bf16_loop(const uint32_t* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = bf16_rne(src[i]);
}

extern "C" void tfr_io_f32_to_bf16(const float* src, uint16_t* dst, size_t n) {
  f32_to_bf16_loop(reinterpret_cast<const uint32_t*>(src), dst, n);
}

// The batch parser behind both entry points.  example_bf16 != nullptr: every record is decoded into a per-thread
// fp32 image of one list (list_size x total width: cache resident) and leaves it rounded to bf16 -- half the bytes
// for the pinned buffer and the host link; context features stay fp32 (a few values per list).
static int parse_batch_impl(int format, const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                                 int32_t list_size, const tfr_io_feature_spec* example_specs,
                                 int32_t n_example, const tfr_io_feature_spec* context_specs,
                                 int32_t n_context, float* example_out, uint16_t* example_bf16, float* context_out,
                                 int32_t* sizes_out, uint8_t* mask_out, int32_t num_threads,
                                 const int32_t* f32_columns = nullptr, int32_t n_f32_columns = 0,
                                 float* f32_out = nullptr) {
  if (B < 0 || list_size <= 0 || n_example <= 0 || !example_specs || (!example_out && !example_bf16) || n_context < 0)
    return TFR_IO_EINVAL;
  if (format != TFR_IO_FORMAT_ELWC && format != TFR_IO_FORMAT_EIE && format != TFR_IO_FORMAT_SEQ &&
      format != TFR_IO_FORMAT_EXAMPLE)
    return TFR_IO_EINVAL;
  if (B > 0 && (!records || !lengths)) return TFR_IO_EINVAL;
  if (n_context > 0 && (!context_specs || !context_out)) return TFR_IO_EINVAL;
  for (int i = 0; i < n_example; ++i) if (!example_specs[i].name || example_specs[i].width < 1) return TFR_IO_EINVAL;
  for (int i = 0; i < n_context; ++i) if (!context_specs[i].name || context_specs[i].width < 1) return TFR_IO_EINVAL;
  const SpecTable ex(example_specs, n_example);
  const SpecTable cx(context_specs, n_context);
  if (n_f32_columns < 0 || (n_f32_columns > 0 && (!f32_columns || !f32_out || !example_bf16))) return TFR_IO_EINVAL;
  for (int i = 0; i < n_f32_columns; ++i) if (f32_columns[i] < 0 || f32_columns[i] >= ex.total) return TFR_IO_EINVAL;
  std::atomic<int> err{0};
  auto work = [&](int lo, int hi) {
    Hints hints;
    const size_t per_list = (size_t)list_size * ex.total;
    std::vector<float> image(example_bf16 ? per_list : 0);
    for (int b = lo; b < hi && err.load(std::memory_order_relaxed) == 0; ++b) {
      float* dst = example_bf16 ? image.data() : example_out + (size_t)b * per_list;
      float* crow = n_context ? context_out + (size_t)b * cx.total : nullptr;
      int32_t* srow = sizes_out ? sizes_out + b : nullptr;
      uint8_t* mrow = mask_out ? mask_out + (size_t)b * list_size : nullptr;
      const SpecTable* cxp = n_context ? &cx : nullptr;
      const uint8_t* rec = records[b]; const size_t len = (size_t)lengths[b];
      int rc;
      if (format == TFR_IO_FORMAT_EXAMPLE) {
        // one tf.Example = one list of one item that also carries the context features (data.py:1348-1395)
        rc = decode_example(rec, len, ex, dst, hints.example, hints.example_tpl);
        if (rc >= 0 && cxp && crow) rc = decode_example(rec, len, *cxp, crow, hints.context, hints.context_tpl);
        for (int i = 1; i < list_size; ++i) ex.fill_defaults(dst + (size_t)i * ex.total);
        if (srow) *srow = 1;
        if (mrow) for (int i = 0; i < list_size; ++i) mrow[i] = i < 1 ? 1 : 0;
      } else {
        rc = format == TFR_IO_FORMAT_ELWC ? decode_elwc(rec, len, list_size, ex, cxp, dst, crow, srow, mrow, hints)
           : format == TFR_IO_FORMAT_EIE  ? decode_eie(rec, len, list_size, ex, cxp, dst, crow, srow, mrow, hints)
                                          : decode_seq(rec, len, list_size, ex, cxp, dst, crow, srow, mrow, hints, false);
      }
      if (rc < 0) { int z = 0; err.compare_exchange_strong(z, rc); }
      else if (example_bf16) {
        tfr_io_f32_to_bf16(dst, example_bf16 + (size_t)b * per_list, per_list);
        float* side = f32_out + (size_t)b * list_size * n_f32_columns;        // labels and the like: unrounded
        for (int i = 0; i < list_size && n_f32_columns > 0; ++i)
          for (int c = 0; c < n_f32_columns; ++c) side[(size_t)i * n_f32_columns + c] = dst[(size_t)i * ex.total + f32_columns[c]];
      }
    }
    g_examples_replayed.fetch_add(hints.example_tpl.replayed + hints.context_tpl.replayed, std::memory_order_relaxed);
    g_examples_walked.fetch_add(hints.example_tpl.walked + hints.context_tpl.walked, std::memory_order_relaxed);
  };
  int T = num_threads > 1 ? std::min(num_threads, std::max(1, B)) : 1;
  if (T <= 1) {
    work(0, B);
  } else {
    std::vector<std::thread> pool;
    const int per = (B + T - 1) / T;
    for (int t = 0; t < T; ++t) pool.emplace_back(work, std::min(B, t * per), std::min(B, (t + 1) * per));
    for (auto& th : pool) th.join();
  }
  return err.load();
}

extern "C" int tfr_io_parse_elwc_batch(const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                                       int32_t list_size, const tfr_io_feature_spec* example_specs,
                                       int32_t n_example, const tfr_io_feature_spec* context_specs,
                                       int32_t n_context, float* example_out, float* context_out,
                                       int32_t* sizes_out, uint8_t* mask_out, int32_t num_threads) {
  if (!example_out) return TFR_IO_EINVAL;
  return parse_batch_impl(TFR_IO_FORMAT_ELWC, records, lengths, B, list_size, example_specs, n_example, context_specs,
                          n_context, example_out, nullptr, context_out, sizes_out, mask_out, num_threads);
}

extern "C" int tfr_io_parse_elwc_batch_bf16(const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                                            int32_t list_size, const tfr_io_feature_spec* example_specs,
                                            int32_t n_example, const tfr_io_feature_spec* context_specs,
                                            int32_t n_context, uint16_t* example_out_bf16, float* context_out,
                                            int32_t* sizes_out, uint8_t* mask_out, int32_t num_threads,
                                            const int32_t* f32_columns, int32_t n_f32_columns, float* f32_out) {
  if (!example_out_bf16) return TFR_IO_EINVAL;
  return parse_batch_impl(TFR_IO_FORMAT_ELWC, records, lengths, B, list_size, example_specs, n_example, context_specs,
                          n_context, nullptr, example_out_bf16, context_out, sizes_out, mask_out, num_threads, f32_columns,
                          n_f32_columns, f32_out);
}

extern "C" int tfr_io_parse_batch(int32_t format, const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                                  int32_t list_size, const tfr_io_feature_spec* example_specs, int32_t n_example,
                                  const tfr_io_feature_spec* context_specs, int32_t n_context, float* example_out,
                                  uint16_t* example_out_bf16, float* context_out, int32_t* sizes_out, uint8_t* mask_out,
                                  int32_t num_threads, const int32_t* f32_columns, int32_t n_f32_columns,
                                  float* f32_out) {
  if ((example_out == nullptr) == (example_out_bf16 == nullptr)) return TFR_IO_EINVAL;    // exactly one of the two
  return parse_batch_impl(format, records, lengths, B, list_size, example_specs, n_example, context_specs, n_context,
                          example_out, example_out_bf16, context_out, sizes_out, mask_out, num_threads, f32_columns,
                          n_f32_columns, f32_out);
}

extern "C" int64_t tfr_io_max_list_size(int32_t format, const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                                        const tfr_io_feature_spec* example_specs, int32_t n_example) {
  if (B < 0 || (B > 0 && (!records || !lengths))) return TFR_IO_EINVAL;
  if (format == TFR_IO_FORMAT_ELWC) return tfr_io_elwc_max_list_size(records, lengths, B);
  if (format == TFR_IO_FORMAT_EXAMPLE) return B > 0 ? 1 : 0;
  if (format != TFR_IO_FORMAT_EIE && format != TFR_IO_FORMAT_SEQ) return TFR_IO_EINVAL;
  if (format == TFR_IO_FORMAT_SEQ) {
    if (n_example <= 0 || !example_specs) return TFR_IO_EINVAL;
    for (int i = 0; i < n_example; ++i) if (!example_specs[i].name || example_specs[i].width < 1) return TFR_IO_EINVAL;
  }
  int64_t best = 0;
  if (format == TFR_IO_FORMAT_EIE) {
    for (int b = 0; b < B; ++b) {
      const int c = count_eie(records[b], (size_t)lengths[b]);
      if (c < 0) return c;
      best = std::max<int64_t>(best, c);
    }
    return best;
  }
  const SpecTable ex(example_specs, n_example);
  Hints hints;
  for (int b = 0; b < B; ++b) {
    const int c = decode_seq(records[b], (size_t)lengths[b], 0, ex, nullptr, nullptr, nullptr, nullptr, nullptr, hints, true);
    if (c < 0) return c;
    best = std::max<int64_t>(best, c);
  }
  return best;
}

extern "C" int64_t tfr_io_libsvm_load(const char* text, size_t nbytes, int32_t list_size, int32_t num_features,
                                      float* features_out, float* labels_out, int64_t* stats_out) {
  if ((!text && nbytes) || list_size <= 0 || num_features <= 0) return TFR_IO_EINVAL;
  if ((features_out == nullptr) != (labels_out == nullptr)) return TFR_IO_EINVAL;
  // Python's float(token) then a float32 store (tf_ranking_libsvm.py:160-181): decimal -> double -> float, which
  // from_chars<double> + a cast reproduces bit for bit; tokens it does not take whole ("+1", "1_0", " inf") go to
  // strtod like before.
  auto to_float = [](const char* b, const char* e, bool& ok) -> float {
    double v = 0.0;
    const auto r = std::from_chars(b, e, v);
    if (r.ec == std::errc() && r.ptr == e) { ok = true; return (float)v; }
    const std::string tmp(b, e);
    char* endp = nullptr;
    v = strtod(tmp.c_str(), &endp);
    ok = endp != tmp.c_str();
    return (float)v;
  };
  std::unordered_map<std::string_view, int64_t> qid_index;   // keys point into `text`
  std::vector<int32_t> ndoc;
  int64_t total = 0, discarded = 0;
  const char* p = text; const char* end = text + nbytes;
  while (p < end) {
    const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    if (!eol) eol = end;
    const char* hash = static_cast<const char*>(memchr(p, '#', (size_t)(eol - p)));
    const char* le = hash ? hash : eol;
    // tokens separated by whitespace
    auto skip_ws = [&](const char* s) { while (s < le && (*s == ' ' || *s == '\t' || *s == '\r')) ++s; return s; };
    auto tok_end = [&](const char* s) { while (s < le && *s != ' ' && *s != '\t' && *s != '\r') ++s; return s; };
    const char* s = skip_ws(p);
    if (s < le) {
      const char* e = tok_end(s);
      bool ok = false;
      const float label = to_float(s, e, ok);
      if (!ok) return TFR_IO_ECORRUPT;
      s = skip_ws(e);
      if (s >= le) return TFR_IO_ECORRUPT;                  // "Ill-formatted line" (:143)
      e = tok_end(s);
      const std::string_view qid(s, (size_t)(e - s));       // the whole token, like the reference (:145)
      auto it = qid_index.find(qid);
      int64_t q;
      if (it == qid_index.end()) {
        q = (int64_t)qid_index.size();
        qid_index.emplace(qid, q);
        ndoc.push_back(0);
      } else {
        q = it->second;
      }
      ++total;
      const int32_t d = ndoc[q]++;
      if (d >= list_size) {
        ++discarded;                                        // keep the first list_size docs only (:176-179)
      } else if (features_out) {
        labels_out[q * list_size + d] = label;
        float* row = features_out + ((size_t)q * list_size + d) * num_features;
        s = skip_ws(e);
        while (s < le) {
          e = tok_end(s);
          const char* colon = static_cast<const char*>(memchr(s, ':', (size_t)(e - s)));
          if (!colon) return TFR_IO_ECORRUPT;
          long fid = 0;
          const auto fr = std::from_chars(s, colon, fid);
          if (fr.ec != std::errc() || fr.ptr != colon) fid = strtol(std::string(s, colon).c_str(), nullptr, 10);
          if (fid < 1 || fid > num_features) return TFR_IO_ESHAPE;   // "Key not found in features" (:181)
          bool vok = false;
          row[fid - 1] = to_float(colon + 1, e, vok);
          if (!vok) return TFR_IO_ECORRUPT;                 // float('abc') raises in the reference too
          s = skip_ws(e);
        }
      }
    }
    p = eol + 1;
  }
  if (stats_out) { stats_out[0] = total; stats_out[1] = discarded; }
  return (int64_t)qid_index.size();
}
