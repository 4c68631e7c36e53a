// gemini::CheetahLinear over libmm_hip: same public surface as the reference class
// (SCI/src/cheetah/cheetah-api.h:16-98) with mm::IOChannel in place of sci::NetIO.
// Callers written against the reference (Conv2DWrapper, MatMul2D, ElemWiseActModelVectorMult in
// SCI/src/library_fixed_uniform_cheetah.cpp; tests/bole_test.cpp) use it the same way.
// Both parties must run this implementation: ciphertexts travel as raw coefficient arrays, not
// in SEAL's serialisation.
#pragma once
#include <memory>
#include <vector>

#include "mm_gemini/core.h"
#include "mm_gemini/hom_conv2d_ss.h"
#include "mm_gemini/hom_bn_ss.h"
#include "mm_gemini/hom_fc_ss.h"
#include "mm_gemini/io.h"

struct mmh_ctx;

namespace gemini {


class CheetahLinear {
 public:
  using ConvMeta = HomConv2DSS::Meta;
  using FCMeta = HomFCSS::Meta;
  using BNMeta = HomBNSS::Meta;

  // party: 1 = ALICE (server, model owner), 2 = BOB (client, key owner)  -- sci::ALICE / sci::BOB
  // With MM_GEMINI_DEVICES="0,1,..." in the environment this constructor takes that device list instead of `device`.
  CheetahLinear(int party, mm::IOChannel *io, uint64_t base_mod, size_t nthreads = 1, int device = 0);
  struct Options {
    size_t nthreads = 1;            // as the reference's argument; the devices do the work
    // devices[0] holds the keys and does all of BOB's work.  ALICE cuts the server side of conv2d / conv2d_nhwc (output
    // filters), fc / fc_in_out (row blocks) and bn (ciphertexts) over all of them, in LaunchWorks' ranges
    // (hom_conv2d_ss.cc:597-607); the bytes on the wire do not depend on the list.  A device may be listed twice.
    // BOLE and bn_direct run on devices[0].
    std::vector<int> devices{0};
    // 8 words; when set, all of the object's randomness (sampler streams and ciphertext seeds) follows from it
    // instead of the OS.  For tests that compare runs; never in production.
    const uint32_t *sampler_key = nullptr;
  };
  CheetahLinear(int party, mm::IOChannel *io, uint64_t base_mod, const Options &options);
  ~CheetahLinear();
  CheetahLinear(const CheetahLinear &) = delete;

  void conv2d(const Tensor<uint64_t> &in_tensor, const std::vector<Tensor<uint64_t>> &filters,
              const ConvMeta &meta, Tensor<uint64_t> &out_tensor) const;
  // conv2d as Conv2DWrapper needs it (library_fixed_uniform_cheetah.cpp:216-300): activations [H][W][CI], filters
  // [FH][FW][CI][CO] (server only) and the result [OH][OW][CO] in the networks' own layouts, raw values (getRingElt is
  // applied on the device); no host-side repacking, no intermediate tensors.  Not in the reference class.
  // expect_oh / expect_ow (when not 0): the output size the caller's array was made for; a layer whose geometry
  // gives another size throws before anything is written.
  void conv2d_nhwc(const uint64_t *input_nhwc, const uint64_t *filters_hwio, const ConvMeta &meta,
                   uint64_t *out_nhwc, int64_t expect_oh = 0, int64_t expect_ow = 0) const;
  // idle device buffers of this object back to the driver (it keeps a size-keyed pool, capped at 2 GiB idle)
  void release_device_buffers() const;
  size_t n_devices() const;  // ALICE: the devices that share a layer; BOB: 1
  void fc(const Tensor<uint64_t> &input_vector, const Tensor<uint64_t> &weight_matrix, const FCMeta &meta,
          Tensor<uint64_t> &out_vec_share) const;
  // the same with the weights as the layer adapter holds them: [meta.weight_shape.cols()][meta.weight_shape.rows()]
  // row-major (input x output), any 64-bit values; transposed and reduced into the ring on the device instead of in
  // a strided host loop (library_fixed_uniform_cheetah.cpp:93-101).  ALICE only reads weights_in_out.
  void fc_in_out(const Tensor<uint64_t> &input_vector, const uint64_t *weights_in_out, const FCMeta &meta,
                 Tensor<uint64_t> &out_vec_share) const;
  void bn(const Tensor<uint64_t> &input_vector, const Tensor<uint64_t> &scale_vector, const BNMeta &meta,
          Tensor<uint64_t> &out_vector) const;
  void BOLE(const Tensor<uint64_t> &input_x_vector, const Tensor<uint64_t> &input_y_vector, const BNMeta &meta,
            Tensor<uint64_t> &out_vector) const;
  // cheetah-api.h: channel-wise scaling of a C x H x W tensor mod base_mod (meta.ishape), coefficient-packed
  void bn_direct(const Tensor<uint64_t> &input_tensor, const Tensor<uint64_t> &scale_vector, const BNMeta &meta,
                 Tensor<uint64_t> &out_tensor) const;

  // Two-party self-check of conv2d (cheetah-api.cpp:206-296): BOB sends both of its shares, ALICE
  // reconstructs input and output, recomputes the convolution in the clear (Tensor::Conv2D,
  // tensor.h:274-362) and prints "HomConv: matches" / "HomConv: failed".  Like the reference it returns
  // false only on a shape error; `mismatches`, when given, receives ALICE's count (the shipped
  // USE_APPROX_RESHARE build may be off by one unit per element).
  bool verify(const Tensor<uint64_t> &in_tensor_share, const std::vector<Tensor<uint64_t>> &filters,
              const ConvMeta &meta, const Tensor<uint64_t> &out_tensor_share, int nbit_precision,
              size_t *mismatches = nullptr) const;

  template <typename T>
  void safe_erase(T *mem, size_t cnt) const {  // cheetah-api.h:63-68
    if (!mem || cnt == 0) return;
    volatile unsigned char *p = reinterpret_cast<volatile unsigned char *>(mem);
    for (size_t i = 0; i < sizeof(T) * cnt; ++i) p[i] = 0;
  }

  uint64_t io_counter() const { return io_ ? io_->counter : 0; }
  int party() const { return party_; }
  uint64_t field_mod() const { return field_mod_; }

 private:
  void fc_any(const Tensor<uint64_t> &input_vector, const uint64_t *wd, bool in_out, const FCMeta &meta,
              Tensor<uint64_t> &out_vec_share) const;
  struct Impl;
  int party_;
  mm::IOChannel *io_;
  uint64_t base_mod_, field_mod_ = 0;
  std::unique_ptr<Impl> impl_;
};

}  // namespace gemini
