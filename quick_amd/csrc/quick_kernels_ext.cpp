// `quick_kernels` as a compiled torch extension over the C ABI of libquick_amd.so -- the MI355X counterpart of the
// reference's csrc/pybind.cpp:5-8 + the host function csrc/gemm_cuda_quick.cu:1456-1517 (same symbol, same positional
// signature, same exceptions, same [M, N] vs [1, M, N] return rule).  Like the reference it takes the buffers of
// WQLinear_QUICK in the REFERENCE's packed order (quick/awq/modules/linear/quick.py:88-150); the MI355X-order copy the HIP
// kernels consume is made once per tensor triple by quick_repack_cuda_to_mi355x and cached (weak references + version
// counters: an in-place rewrite or a dead tensor drops the entry).  No arithmetic here: shape rules, allocation, the cache.
//
// Built by quick_amd/build_ext.py (torch.utils.cpp_extension, in-tree); tests/test_gemm_gpu.py builds and calls it.
#include <torch/extension.h>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // PyTorch-ROCm tensors carry the device type "cuda": the guard and the
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>      // stream accessor that accept it are these two

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "quick_amd.h"

namespace {

struct Entry {
  std::vector<c10::weak_intrusive_ptr<c10::TensorImpl>> ref;
  uint32_t version[3];
  torch::Tensor packed[3];
};
using Key = std::tuple<const void*, const void*, const void*>;
std::map<Key, Entry> g_cache;
std::map<std::pair<int, void*>, torch::Tensor> g_workspace;  // (device, stream) -> zero-filled scratch, handed back zeroed by the library
std::mutex g_mutex;

uint32_t version_of(const torch::Tensor& t) { return t.is_inference() ? 0u : t._version(); }

void check(const torch::Tensor& t, c10::ScalarType dtype, const char* name) {
  // the reference takes data_ptr<T>(): a wrong dtype is a c10::Error (RuntimeError in Python)
  TORCH_CHECK(t.scalar_type() == dtype, "expected scalar type ", dtype, " for ", name, " but found ", t.scalar_type());
  TORCH_CHECK(t.is_cuda(), name, " must be a GPU tensor: the W4A16 GEMM has no CPU implementation");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

[[noreturn]] void raise(int rc) {
  const std::string msg = quick_amd_last_error();
  if (rc == QUICK_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);  // -> ValueError, as the reference's host function
  if (rc == QUICK_ERR_UNSUPPORTED) {   // NotImplementedError, the same type the ctypes shim raises (quick_amd/kernels.py:_raise)
    PyErr_SetString(PyExc_NotImplementedError, ("unsupported on MI355X: " + msg).c_str());
    throw py::error_already_set();
  }
  TORCH_CHECK(false, msg);
}

}  // namespace

torch::Tensor gemm_forward_cuda_quick(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor scaling_factors, torch::Tensor zeros,
                                      int split_k_iters) {
  if (split_k_iters < 1) throw std::invalid_argument("split_k_iters must be >= 1");
  check(in_feats, torch::kHalf, "in_feats");
  check(kernel, torch::kInt, "kernel");
  check(scaling_factors, torch::kHalf, "scaling_factors");
  check(zeros, torch::kInt, "zeros");
  TORCH_CHECK(in_feats.dim() == 2, "in_feats must be 2-D [M, K]");
  const int M = (int)in_feats.size(0), K = (int)in_feats.size(1);
  const int N = (int)(kernel.size(1) / 4 * 8);                 // gemm_cuda_quick.cu:1468
  if (kernel.size(0) * 4 != K) throw std::invalid_argument("kernel and in_feats disagree on the number of input channels");
  const int G = K / (int)scaling_factors.size(0);              // gemm_cuda_quick.cu:1477
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(in_feats.device());   // OptionalCUDAGuard, gemm_cuda_quick.cu:1465
  hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  // in_features % 128 != 0 (the reference takes % 32): the MI355X-order copy is padded along K, the activations with zeros
  const int padded = (K % 128 != 0) ? quick_padded_in_features(K, G) : 0;
  const int Kp = padded > 0 ? padded : K;
  {  // the reference's shape errors, before any repack
    char text[8];
    const int rc = quick_w4a16_plan_describe(M > 0 ? M : 1, Kp, N, G, QUICK_KERNEL_AUTO, 0, text, sizeof(text));
    if (rc != QUICK_OK) raise(rc);
  }
  if (Kp != K) in_feats = torch::constant_pad_nd(in_feats, {0, Kp - K}, 0);
  torch::Tensor packed[3];
  torch::Tensor ws;
  size_t ws_bytes = 0;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    const Key key{kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr()};
    const torch::Tensor src[3] = {kernel, scaling_factors, zeros};
    auto it = g_cache.find(key);
    bool hit = it != g_cache.end();
    for (int i = 0; hit && i < 3; ++i) {
      auto alive = it->second.ref[i].lock();
      hit = alive && alive.get() == src[i].unsafeGetTensorImpl() && it->second.version[i] == version_of(src[i]);
    }
    if (!hit) {
      Entry e;
      for (int i = 0; i < 3; ++i) {
        e.ref.emplace_back(src[i].getIntrusivePtr());
        e.version[i] = version_of(src[i]);
      }
      e.packed[0] = torch::empty({Kp / 4, N / 2}, kernel.options());
      e.packed[1] = torch::empty({Kp / G, 2 * N}, scaling_factors.options());
      e.packed[2] = torch::empty({Kp / G, N / 4}, zeros.options());
      const int rc = (Kp != K ? quick_repack_cuda_to_mi355x_padded : quick_repack_cuda_to_mi355x)(
          kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), e.packed[0].data_ptr(), e.packed[1].data_ptr(),
          e.packed[2].data_ptr(), K, N, G, stream);
      if (rc != QUICK_OK) raise(rc);
      // The copy is cached for every later caller, whatever stream it is on: finish the repack once, here, so that a first use
      // from another stream cannot read a half-written copy (not inside a stream capture, where synchronising is not allowed --
      // there the repack is part of the captured work and ordered with it).
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap == hipStreamCaptureStatusNone) (void)hipStreamSynchronize(stream);
      (void)hipGetLastError();
      for (auto d = g_cache.begin(); d != g_cache.end();)  // drop entries whose tensors died
        d = (d->second.ref[0].expired() || d->second.ref[1].expired() || d->second.ref[2].expired()) ? g_cache.erase(d) : std::next(d);
      it = g_cache.insert_or_assign(key, std::move(e)).first;
    }
    for (int i = 0; i < 3; ++i) packed[i] = it->second.packed[i];
    ws_bytes = M > 0 ? quick_w4a16_workspace_bytes(M, Kp, N, G, split_k_iters) : 0;
    if (ws_bytes) {  // ZERO-FILLED on first use (the arrival counters of the in-kernel split-K reduction), reused afterwards
      auto& slot = g_workspace[{(int)in_feats.get_device(), (void*)stream}];
      if (!slot.defined() || (size_t)slot.numel() < ws_bytes)
        slot = torch::zeros({(int64_t)std::max<size_t>(ws_bytes, 1 << 20)}, in_feats.options().dtype(torch::kUInt8));
      ws = slot;
    }
  }
  auto y = torch::empty({M, N}, in_feats.options());
  if (M > 0) {
    const int rc = quick_w4a16_gemm_f16(in_feats.data_ptr(), packed[0].data_ptr(), packed[1].data_ptr(), packed[2].data_ptr(), y.data_ptr(),
                                        ws_bytes ? ws.data_ptr() : nullptr, ws_bytes ? (size_t)ws.numel() : 0, M, Kp, N, G, split_k_iters, stream);
    if (rc != QUICK_OK) raise(rc);
  }
  return split_k_iters > 1 ? y : y.unsqueeze(0);               // gemm_cuda_quick.cu:1515-1516
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("gemm_forward_cuda_quick", &gemm_forward_cuda_quick, "QUICK AWQ GEMM kernel (MI355X)."); }
