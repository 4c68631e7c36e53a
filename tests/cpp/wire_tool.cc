// Command-line face of include/seal/wire.h for tests/test_seal_wire.py (host code only: g++, no device library).
//   wire_tool blake2b  OUTLEN HEXIN [HEXKEY]
//   wire_tool blake2xb OUTLEN HEXIN HEXKEY
//   wire_tool parms    SCHEME N NSPECIAL T Q...
//   wire_tool uniform  HEXSEED N Q...              words of sample_poly_uniform, hex, one per line
//   wire_tool save     SEEDED NTT SIZE N HEXSEED T Q...   a ciphertext with data[i] = f(i) mod q_l, SEAL bytes as hex
//   wire_tool load     < hex on stdin              prints the members and a checksum of the words (c1 expanded if seeded)
//   wire_tool inflate  < hex on stdin              Zstandard frames -> content, hex
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "seal/wire.h"

using namespace seal::wire;

static std::vector<std::uint8_t> unhex(const std::string &s) {
  std::vector<std::uint8_t> b;
  for (std::size_t i = 0; i + 1 < s.size(); i += 2) b.push_back((std::uint8_t)std::stoul(s.substr(i, 2), nullptr, 16));
  return b;
}
static void hex(const void *p, std::size_t n) {
  for (std::size_t i = 0; i < n; ++i) std::printf("%02x", static_cast<const std::uint8_t *>(p)[i]);
  std::printf("\n");
}
static std::uint64_t fill_word(std::size_t i, std::uint64_t q) { return (0x9E3779B97F4A7C15ull * (i + 1) + 12345) % q; }

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  try {
    if (cmd == "blake2b" || cmd == "blake2xb") {
      const std::size_t outlen = std::stoul(argv[2]);
      const auto in = unhex(argv[3]);
      const auto key = argc > 4 ? unhex(argv[4]) : std::vector<std::uint8_t>();
      std::vector<std::uint8_t> out(outlen);
      if (cmd == "blake2b")
        blake2b(out.data(), outlen, in.data(), in.size(), key.data(), key.size());
      else
        blake2xb(out.data(), outlen, in.data(), in.size(), key.data(), key.size());
      hex(out.data(), outlen);
    } else if (cmd == "parms") {
      std::vector<std::uint64_t> q;
      for (int i = 6; i < argc; ++i) q.push_back(std::stoull(argv[i]));
      const auto id = parms_id_hash((std::uint8_t)std::stoul(argv[2]), std::stoull(argv[3]), std::stoull(argv[4]), q,
                                    std::stoull(argv[5]));
      for (auto w : id) std::printf("%016" PRIx64 "\n", w);
    } else if (cmd == "uniform") {
      const auto seedv = unhex(argv[2]);
      std::uint8_t seed[kSeedBytes];
      std::memcpy(seed, seedv.data(), kSeedBytes);
      const std::size_t N = std::stoul(argv[3]);
      std::vector<std::uint64_t> q;
      for (int i = 4; i < argc; ++i) q.push_back(std::stoull(argv[i]));
      std::vector<std::uint64_t> dst(q.size() * N);
      Blake2xbPRNG prng(seed);
      sample_poly_uniform(prng, q, N, dst.data());
      for (auto w : dst) std::printf("%016" PRIx64 "\n", w);
    } else if (cmd == "save") {
      const bool seeded = std::stoi(argv[2]) != 0;
      CtMeta m;
      m.is_ntt_form = std::stoi(argv[3]) != 0;
      m.size = std::stoull(argv[4]), m.poly_modulus_degree = std::stoull(argv[5]);
      const auto seedv = unhex(argv[6]);
      std::uint8_t seed[kSeedBytes];
      std::memcpy(seed, seedv.data(), kSeedBytes);
      const std::uint64_t t = std::stoull(argv[7]);
      std::vector<std::uint64_t> q;
      for (int i = 8; i < argc; ++i) q.push_back(std::stoull(argv[i]));
      m.coeff_modulus_size = q.size();
      m.parms_id = parms_id_hash(1, m.poly_modulus_degree, 0, q, t);
      const std::size_t N = (std::size_t)m.poly_modulus_degree, L = q.size();
      std::vector<std::uint64_t> data((std::size_t)m.size * L * N);
      for (std::size_t k = 0; k < m.size; ++k)
        for (std::size_t l = 0; l < L; ++l)
          for (std::size_t i = 0; i < N; ++i) data[(k * L + l) * N + i] = fill_word((k * L + l) * N + i, q[l]);
      const std::string bytes = save_ciphertext(m, data.data(), seeded ? seed : nullptr);
      hex(bytes.data(), bytes.size());
    } else if (cmd == "load") {
      std::string h;
      std::cin >> h;
      const auto bytes = unhex(h);
      CtMeta m;
      std::vector<std::uint64_t> data;
      bool seeded = false;
      std::uint8_t seed[kSeedBytes] = {0};
      std::size_t consumed = 0;
      const std::size_t words = load_ciphertext(bytes.data(), bytes.size(), m, data, seeded, seed, &consumed);
      std::printf("parms_id %016" PRIx64 " %016" PRIx64 " %016" PRIx64 " %016" PRIx64 "\n", m.parms_id[0], m.parms_id[1],
                  m.parms_id[2], m.parms_id[3]);
      std::printf("ntt %d size %" PRIu64 " N %" PRIu64 " L %" PRIu64 " cf %" PRIu64 " scale %.17g seeded %d words %zu consumed %zu\n",
                  (int)m.is_ntt_form, m.size, m.poly_modulus_degree, m.coeff_modulus_size, m.correction_factor, m.scale,
                  (int)seeded, words, consumed);
      if (seeded) hex(seed, kSeedBytes);
      std::uint64_t sum = 0;
      for (std::size_t i = 0; i < words; ++i) sum = sum * 0x100000001B3ull + data[i];
      std::printf("checksum %016" PRIx64 "\n", sum);
    } else if (cmd == "inflate") {   // hex of Zstandard frames on stdin -> hex of the content
      std::string h;
      std::cin >> h;
      const auto bytes = unhex(h);
      const auto out = zstd::decompress(bytes.data(), bytes.size());
      hex(out.data(), out.size());
    } else {
      return 2;
    }
  } catch (const std::exception &e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
  return 0;
}
