// The two host-side services the path needs from its caller, as abstract interfaces:
//  * a byte channel with the surface CheetahLinear uses on sci::NetIO
//    (SCI/src/utils/net_io_channel.h:63-163: send_data / recv_data / flush / counter), and
//  * the correlated-OT calls Truncation makes on otpack->iknp_straight
//    (SCI/src/OT/ferret/silent_ot.h:85-242).
// Production code adapts sci::NetIO and SilentOT to these in a dozen lines; the in-process
// implementations below are test doubles (LoopbackIO; IdealCOT in the style of
// SCI/src/OT/ideal.h:106-156).
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <vector>

namespace mm {

class IOChannel {
 public:
  virtual ~IOChannel() = default;
  virtual void send_data(const void *data, size_t nbytes) = 0;
  virtual void recv_data(void *data, size_t nbytes) = 0;
  virtual void flush() {}
  uint64_t counter = 0;  // bytes sent, as sci::IOChannel::counter (io_channel.h:41-45)
};

// One direction of an in-process byte pipe.  A receiver that is already waiting gets the sender's bytes copied
// straight into its buffer (one copy, like a socket read into user memory); otherwise the message is queued in a
// chunk (recycled: a fresh 100-MiB vector would fault its pages in on every layer) and copied out later.
class BytePipe {
 public:
  void push(const void *p, size_t n) {
    if (n == 0) return;
    const uint8_t *b = static_cast<const uint8_t *>(p);
    std::unique_lock<std::mutex> g(m_);
    if (want_ && q_.empty()) {  // direct hand-over into the waiting receiver's buffer
      const size_t take = std::min(n, want_);
      timed_copy(dst_, b, take);
      dst_ += take, want_ -= take, b += take, n -= take;
      if (want_ == 0) cv_.notify_all();
      if (n == 0) return;
    }
    std::vector<uint8_t> chunk;
    if (!spare_.empty()) {
      chunk = std::move(spare_.back());
      spare_.pop_back();
    }
    const auto t0 = std::chrono::steady_clock::now();
    chunk.assign(b, b + n);
    copy_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    bytes_ += n;
    q_.push_back(std::move(chunk));
    cv_.notify_all();
  }
  void pop(void *p, size_t n) {
    uint8_t *dst = static_cast<uint8_t *>(p);
    std::unique_lock<std::mutex> g(m_);
    while (n > 0) {
      if (q_.empty()) {  // nothing queued: post the buffer and let the sender fill it
        dst_ = dst, want_ = n;
        cv_.wait(g, [&] { return want_ == 0 || !q_.empty(); });
        const size_t got = n - want_;
        dst += got, n -= got;
        want_ = 0, dst_ = nullptr;
        continue;
      }
      std::vector<uint8_t> &front = q_.front();
      const size_t take = std::min(n, front.size() - head_);
      timed_copy(dst, front.data() + head_, take);
      dst += take, n -= take, head_ += take, bytes_ -= take;
      if (head_ == front.size()) {
        if (spare_.size() < 4) spare_.push_back(std::move(front));
        q_.pop_front();
        head_ = 0;
      }
    }
  }
  // nanoseconds this pipe has spent copying bytes (the stand-in for the network between the parties), so that a
  // measurement can tell the transport from the work on either side
  uint64_t copy_ns() {
    std::lock_guard<std::mutex> g(m_);
    return copy_ns_;
  }

 private:
  void timed_copy(void *d, const void *s, size_t n) {
    const auto t0 = std::chrono::steady_clock::now();
    std::memcpy(d, s, n);
    copy_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  uint64_t copy_ns_ = 0;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::vector<uint8_t>> q_;
  std::vector<std::vector<uint8_t>> spare_;  // emptied chunks kept for reuse
  size_t head_ = 0, bytes_ = 0;              // read offset inside the front chunk, bytes queued
  uint8_t *dst_ = nullptr;                   // a receiver's posted buffer and what it still needs
  size_t want_ = 0;
};

class LoopbackIO : public IOChannel {
 public:
  LoopbackIO(std::shared_ptr<BytePipe> tx, std::shared_ptr<BytePipe> rx) : tx_(tx), rx_(rx) {}
  static std::pair<std::unique_ptr<LoopbackIO>, std::unique_ptr<LoopbackIO>> MakePair() {
    auto a = std::make_shared<BytePipe>(), b = std::make_shared<BytePipe>();
    return {std::make_unique<LoopbackIO>(a, b), std::make_unique<LoopbackIO>(b, a)};
  }
  void send_data(const void *d, size_t n) override {
    tx_->push(d, n);
    counter += n;
  }
  void recv_data(void *d, size_t n) override { rx_->pop(d, n); }
  uint64_t transport_ns() const { return tx_->copy_ns() + rx_->copy_ns(); }  // both directions of the pair

 private:
  std::shared_ptr<BytePipe> tx_, rx_;
};

// Correlated OT as Truncation uses it: sender gets random x, receiver gets x + b*corr.
class COT {
 public:
  virtual ~COT() = default;
  virtual void send_cot(uint64_t *data0, const uint64_t *corr, int n, int l) = 0;
  virtual void recv_cot(uint64_t *data, const bool *b, int n, int l) = 0;
  virtual void send_cot_prime(uint64_t *data0, const uint64_t *corr, int n, uint64_t p) = 0;
  virtual void recv_cot_prime(uint64_t *data, const bool *b, int n, uint64_t p) = 0;
  virtual void flush() {}
};

// Ideal functionality over a shared in-process mailbox (sender posts (x, corr), receiver combines).
class IdealCOT : public COT {
 public:
  struct Mailbox {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::pair<std::vector<uint64_t>, std::vector<uint64_t>>> q;
  };
  IdealCOT(std::shared_ptr<Mailbox> box, uint64_t seed) : box_(box), rng_(seed) {}
  void send_cot(uint64_t *d, const uint64_t *corr, int n, int l) override { send(d, corr, n, l, 0); }
  void recv_cot(uint64_t *d, const bool *b, int n, int l) override { recv(d, b, n, l, 0); }
  void send_cot_prime(uint64_t *d, const uint64_t *corr, int n, uint64_t p) override { send(d, corr, n, 0, p); }
  void recv_cot_prime(uint64_t *d, const bool *b, int n, uint64_t p) override { recv(d, b, n, 0, p); }

 private:
  static uint64_t mask(int l) { return l >= 64 ? ~uint64_t(0) : (uint64_t(1) << l) - 1; }
  void send(uint64_t *d, const uint64_t *corr, int n, int l, uint64_t p) {
    std::vector<uint64_t> x(n), c(corr, corr + n);
    for (int i = 0; i < n; ++i) x[i] = p ? rng_() % p : rng_() & mask(l);
    std::copy(x.begin(), x.end(), d);
    std::lock_guard<std::mutex> g(box_->m);
    box_->q.emplace_back(std::move(x), std::move(c));
    box_->cv.notify_all();
  }
  void recv(uint64_t *d, const bool *b, int n, int l, uint64_t p) {
    std::unique_lock<std::mutex> g(box_->m);
    box_->cv.wait(g, [&] { return !box_->q.empty(); });
    auto msg = std::move(box_->q.front());
    box_->q.pop_front();
    g.unlock();
    for (int i = 0; i < n; ++i) {
      if (p) d[i] = (uint64_t)(((unsigned __int128)msg.first[i] + (b[i] ? msg.second[i] % p : 0)) % p);
      else d[i] = (msg.first[i] + (b[i] ? msg.second[i] : 0)) & mask(l);
    }
  }
  std::shared_ptr<Mailbox> box_;
  std::mt19937_64 rng_;
};

}  // namespace mm
