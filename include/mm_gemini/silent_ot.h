// mm::SilentOT: the four correlated-OT calls of SCI's SilentOT (SCI/src/OT/ferret/silent_ot.h:85-242) as a mm::COT
// whose post-processing -- MITCCRH hash, mask or mod p, correction, wire packing -- runs on the device
// (mmh_cot_send / mmh_cot_recv), one launch per call.  The random correlated OT underneath (Ferret: Delta,
// send_ot_rcm_cc, recv_ot_rcm_cc; silent_ot.h:89,:131,:173,:211) and the channel stay on the host behind the two
// interfaces below; a production build adapts emp-ot's FerretCOT and sci::NetIO to them.
#pragma once
#include <cstdint>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <vector>

#include "mm_gemini/io.h"

namespace mm {
struct Block {  // emp::block in memory order
  uint64_t lo, hi;
};

class RandomCOT {
 public:
  virtual ~RandomCOT() = default;
  virtual Block Delta() const = 0;                                          // the sender's global offset
  virtual void send_ot_rcm_cc(Block *data0, int64_t n) = 0;                 // silent_ot.h: ferret->send_cot
  virtual void recv_ot_rcm_cc(Block *data, const bool *b, int64_t n) = 0;   // data = data0 ^ b * Delta
};

// Ideal functionality over a shared in-process mailbox: test double (pattern SCI/src/OT/ideal.h).
class IdealFerret : public RandomCOT {
 public:
  struct Mailbox {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::vector<Block>> q;
  };
  IdealFerret(std::shared_ptr<Mailbox> box, uint64_t seed) : box_(box), rng_(seed) {
    std::mt19937_64 d(0x4D264D);
    delta_ = Block{d() | 1, d()};
  }
  Block Delta() const override { return delta_; }
  void send_ot_rcm_cc(Block *data0, int64_t n) override {
    std::vector<Block> x(n);
    for (auto &b : x) b = Block{rng_(), rng_()};
    std::copy(x.begin(), x.end(), data0);
    std::lock_guard<std::mutex> g(box_->m);
    box_->q.push_back(std::move(x));
    box_->cv.notify_all();
  }
  void recv_ot_rcm_cc(Block *data, const bool *b, int64_t n) override {
    std::unique_lock<std::mutex> g(box_->m);
    box_->cv.wait(g, [&] { return !box_->q.empty(); });
    std::vector<Block> x = std::move(box_->q.front());
    box_->q.pop_front();
    g.unlock();
    for (int64_t i = 0; i < n; ++i)
      data[i] = b[i] ? Block{x[i].lo ^ delta_.lo, x[i].hi ^ delta_.hi} : x[i];
  }

 private:
  std::shared_ptr<Mailbox> box_;
  std::mt19937_64 rng_;
  Block delta_;
};

class SilentOT : public COT {
 public:
  // The MITCCRH start point S of every call travels in the clear and must be unpredictable (silent_ot.h:91-95 draws
  // it from emp-tool's AES-based PRG, itself seeded from the OS): 16 bytes of /dev/urandom per call.  A test may ask
  // for a repeatable stream instead by passing a seed.
  SilentOT(RandomCOT *ferret, IOChannel *io) : ferret_(ferret), io_(io) {}
  SilentOT(RandomCOT *ferret, IOChannel *io, uint64_t test_seed)
      : ferret_(ferret), io_(io), seeded_(true), prg_(test_seed) {}
  // silent_ot.h:85, :129, :168, :209 (sci::OT<T> forwards send_cot .. recv_cot_prime to them, ot.h:80-93)
  void send_ot_cam_cc(uint64_t *data0, const uint64_t *corr, int64_t length, int l) { send(data0, corr, length, l, 0); }
  void send_ot_cam_cc_prime(uint64_t *data0, const uint64_t *corr, int64_t length, uint64_t p) {
    send(data0, corr, length, 0, p);
  }
  void recv_ot_cam_cc(uint64_t *data, const bool *b, int64_t length, int l) { recv(data, b, length, l, 0); }
  void recv_ot_cam_cc_prime(uint64_t *data, const bool *b, int64_t length, uint64_t p) { recv(data, b, length, 0, p); }

  void send_cot(uint64_t *data0, const uint64_t *corr, int n, int l) override { send(data0, corr, n, l, 0); }
  void recv_cot(uint64_t *data, const bool *b, int n, int l) override { recv(data, b, n, l, 0); }
  void send_cot_prime(uint64_t *data0, const uint64_t *corr, int n, uint64_t p) override { send(data0, corr, n, 0, p); }
  void recv_cot_prime(uint64_t *data, const bool *b, int n, uint64_t p) override { recv(data, b, n, 0, p); }
  void flush() override { io_->flush(); }

  uint64_t gid = 0;  // MITCCRH's key counter: runs on across calls (setS does not reset it)

 private:
  void send(uint64_t *data0, const uint64_t *corr, int64_t n, int l, uint64_t p);
  void recv(uint64_t *data, const bool *b, int64_t n, int l, uint64_t p);
  void fresh_seed(uint64_t seed[2]);
  RandomCOT *ferret_;
  IOChannel *io_;
  bool seeded_ = false;
  std::mt19937_64 prg_;  // test streams only
};
}  // namespace mm
