// Mutation fuzzer for libtfr_io's record parsers (tfr_io_parse_batch / tfr_io_max_list_size / tfr_io_tfrecord_index /
// tfr_io_libsvm_load): built WITH the library source under -fsanitize=address,undefined, fed valid records of the four
// formats with random byte flips, truncations, splices and length-field edits.  The parsers may return any error code;
// they must never read or write out of bounds.  tests/test_data_cpu.py builds and runs it for a fixed number of rounds.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -pthread -I include \
//       tools/io_fuzz.cpp ranking_amd/csrc/tfr_io.cpp -o /tmp/io_fuzz && /tmp/io_fuzz 20000
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/tfr_io.h"

namespace {
uint64_t rng_state = 0x9e3779b97f4a7c15ull;
uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

void varint(std::string& o, uint64_t v) { while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; } o.push_back((char)v); }
std::string ld(int field, const std::string& p) { std::string o; varint(o, ((uint64_t)field << 3) | 2); varint(o, p.size()); return o + p; }
std::string float_feature(int n, bool packed) {
  std::string inner;
  if (packed) { std::string pl; for (int i = 0; i < n; ++i) { float f = (float)(rnd() % 1000) / 7.0f; pl.append((const char*)&f, 4); } inner = ld(1, pl); }
  else for (int i = 0; i < n; ++i) { float f = (float)(rnd() % 1000) / 7.0f; varint(inner, (1 << 3) | 5); inner.append((const char*)&f, 4); }
  return ld(2, inner);
}
std::string int_feature(int n, bool packed) {
  std::string inner;
  if (packed) { std::string pl; for (int i = 0; i < n; ++i) varint(pl, rnd() % 100000); inner = ld(1, pl); }
  else for (int i = 0; i < n; ++i) { varint(inner, 1 << 3); varint(inner, rnd() % 100000); }
  return ld(3, inner);
}
std::string bytes_feature(int n) { std::string inner; for (int i = 0; i < n; ++i) inner += ld(1, std::string((size_t)(rnd() % 6), 'x')); return ld(1, inner); }
std::string entry(const std::string& key, const std::string& feature) { return ld(1, ld(1, key) + ld(2, feature)); }
std::string features_payload(bool packed) {
  std::string e;
  if (rnd() % 8) e += entry("a", float_feature(1, packed));
  if (rnd() % 8) e += entry("b", float_feature(3, packed));
  if (rnd() % 4) e += entry("c", int_feature(1, packed));
  if (rnd() % 3 == 0) e += entry("tok", bytes_feature(2));
  if (rnd() % 3 == 0) e += entry("other", float_feature(2, packed));
  return e;
}
std::string example(bool packed) { return ld(1, features_payload(packed)); }
std::string elwc(bool packed) { std::string o; const int n = (int)(rnd() % 7); for (int i = 0; i < n; ++i) o += ld(1, example(packed)); if (rnd() % 4) o += ld(2, example(packed)); return o; }
std::string eie(bool packed) {
  std::string exs; const int n = (int)(rnd() % 7);
  for (int i = 0; i < n; ++i) exs += ld(1, example(packed));
  std::string e = entry("serialized_examples", ld(1, exs));
  if (rnd() % 8) e += entry("serialized_context", ld(1, ld(1, example(packed))));
  return ld(1, e);
}
std::string seq(bool packed) {
  std::string lists;
  const char* names[] = {"a", "b", "c", "zz"};
  const int widths[] = {1, 3, 1, 2};
  for (int k = 0; k < 4; ++k) {
    if (rnd() % 5 == 0) continue;
    std::string fl; const int frames = (int)(rnd() % 6);
    for (int t = 0; t < frames; ++t) fl += ld(1, rnd() % 16 == 0 ? std::string() : (k == 2 ? int_feature(widths[k], packed) : float_feature(widths[k], packed)));
    lists += ld(1, ld(1, names[k]) + ld(2, fl));
  }
  std::string o;
  if (rnd() % 4) o += ld(1, features_payload(packed));
  return o + ld(2, lists);
}

void mutate(std::string& r) {
  const int kind = (int)(rnd() % 6);
  if (r.empty()) return;
  if (kind == 0) r.resize((size_t)(rnd() % r.size()));                               // truncate
  else if (kind == 1) { for (int i = 0, n = 1 + (int)(rnd() % 4); i < n; ++i) r[(size_t)(rnd() % r.size())] ^= (char)(1 << (rnd() % 8)); }
  else if (kind == 2) { for (int i = 0, n = 1 + (int)(rnd() % 3); i < n; ++i) r[(size_t)(rnd() % r.size())] = (char)rnd(); }
  else if (kind == 3) { const size_t a = (size_t)(rnd() % r.size()); r.insert(a, r.substr((size_t)(rnd() % r.size()), (size_t)(rnd() % 9))); }
  else if (kind == 4) { const size_t a = (size_t)(rnd() % r.size()); r.erase(a, (size_t)(rnd() % 5)); }
  // kind 5: leave valid
}
}  // namespace

int main(int argc, char** argv) {
  const long rounds = argc > 1 ? atol(argv[1]) : 20000;
  if (argc > 2) rng_state ^= (uint64_t)atoll(argv[2]) * 0x100000001b3ull;
  tfr_io_feature_spec ex[3] = {{"a", 1, 0.5f}, {"b", 3, -2.0f}, {"c", 1, 7.0f}};
  tfr_io_feature_spec cx[2] = {{"a", 1, 0.0f}, {"c", 1, 1.0f}};
  long ok = 0, bad = 0;
  for (long it = 0; it < rounds; ++it) {
    const int fmt = (int)(rnd() % 4);
    const bool packed = rnd() % 2;
    const int B = 1 + (int)(rnd() % 4);
    std::vector<std::string> recs;
    for (int b = 0; b < B; ++b) {
      std::string r = fmt == 0 ? elwc(packed) : fmt == 1 ? eie(packed) : fmt == 2 ? seq(packed) : example(packed);
      if (rnd() % 3) mutate(r);
      if (rnd() % 5 == 0) mutate(r);
      recs.push_back(r);
    }
    std::vector<const uint8_t*> ptrs; std::vector<uint64_t> lens;
    std::vector<std::vector<uint8_t>> exact;                  // heap copies of the exact size: an over-read trips ASAN
    for (auto& r : recs) { exact.emplace_back(r.begin(), r.end()); }
    for (auto& e : exact) { ptrs.push_back(e.data()); lens.push_back(e.size()); }
    const int L = 1 + (int)(rnd() % 5);
    const int threads = 1 + (int)(rnd() % 3);
    std::vector<float> out((size_t)B * L * 5), cout_((size_t)B * 2), side((size_t)B * L * 2);
    std::vector<uint16_t> out16((size_t)B * L * 5);
    std::vector<int32_t> sizes(B); std::vector<uint8_t> mask((size_t)B * L);
    const int32_t cols[2] = {0, 4};
    int rc;
    if (rnd() % 2) rc = tfr_io_parse_batch(fmt, ptrs.data(), lens.data(), B, L, ex, 3, cx, 2, out.data(), nullptr, cout_.data(), sizes.data(), mask.data(), threads, nullptr, 0, nullptr);
    else rc = tfr_io_parse_batch(fmt, ptrs.data(), lens.data(), B, L, ex, 3, cx, 2, nullptr, out16.data(), cout_.data(), sizes.data(), mask.data(), threads, cols, 2, side.data());
    (rc == 0 ? ok : bad)++;
    (void)tfr_io_max_list_size(fmt, ptrs.data(), lens.data(), B, ex, 3);
    if (it % 7 == 0) {                                        // TFRecord framing and the LibSVM loader on the same bytes
      std::vector<uint64_t> off(8), len(8);
      (void)tfr_io_tfrecord_index(exact[0].data(), exact[0].size(), (int)(rnd() % 2), off.data(), len.data(), 8);
      std::string text = "1 qid:1 1:0.5 3:2\n0 qid:1 2:1e3 # c\n2 qid:2 1:nan 136:1\n";
      if (rnd() % 2) mutate(text);
      std::vector<float> f((size_t)8 * 3 * 136), l((size_t)8 * 3); int64_t stats[2];
      std::vector<char> t(text.begin(), text.end());
      const int64_t q = tfr_io_libsvm_load(t.data(), t.size(), 3, 136, nullptr, nullptr, stats);
      if (q >= 0 && q <= 8) (void)tfr_io_libsvm_load(t.data(), t.size(), 3, 136, f.data(), l.data(), stats);
    }
  }
  printf("io_fuzz: %ld rounds, %ld batches parsed, %ld rejected\n", rounds, ok, bad);
  return 0;
}
