"""``WQLinear_QUICK`` for MI355X: same constructor, ``from_linear``, ``forward``, buffer names, shapes and
dtypes as the reference operator (quick/awq/modules/linear/quick.py:35-171), so the callers in the
``quick.awq`` fork (quantizer.py:158-174, models/base.py:417-436, utils/fused_utils.py:97-117) work
unchanged.

Layout contract
---------------
* ``state_dict()`` / ``load_state_dict()`` always speak the reference's packed order ("cuda order"):
  checkpoints are interchangeable with the reference in both directions.
* In memory, after :meth:`prepare` (run lazily by the first ``forward``), the three buffers hold the
  same bits permuted into MI355X order -- the order in which one 16-byte load per lane is the MFMA
  A-operand of four k-steps.  Whether the buffers are currently prepared is tracked by the identity and
  version of their storage, so anything that rewrites them from outside (``load_state_dict``,
  accelerate's ``set_module_tensor_to_device``, plain attribute assignment as in ``fuse_qkv_quick``)
  automatically marks them as reference order again.
"""
import torch
import torch.nn as nn

from . import kernels, packing


class _Layout:
    """Which order the packed buffers of a ``WQLinear_QUICK`` hold.  ``prepared`` is the fact; ``key`` (identity and
    version of the three storages) is how a rewrite from outside is noticed.  Copies and pickles of a module carry the
    fact and re-key themselves on first use -- their buffers are new storages holding the same bits."""
    __slots__ = ("prepared", "key")

    def __init__(self, prepared=False, key=None):
        self.prepared, self.key = prepared, key

    def __deepcopy__(self, memo):
        return _Layout(self.prepared, None)

    def __getstate__(self):
        return {"prepared": self.prepared}

    def __setstate__(self, state):
        self.prepared, self.key = state["prepared"], None


class WQLinear_QUICK(nn.Module):
    def __init__(self, w_bit, group_size, in_features, out_features, bias, dev, k_split_1=2, k_split_2=8):
        super().__init__()
        if w_bit not in [4]:
            raise NotImplementedError("Only 4-bit are supported for now.")
        self.in_features = in_features
        self.out_features = out_features
        self.w_bit = w_bit
        self.group_size = group_size if group_size != -1 else in_features
        self.k_split_1 = k_split_1          # kept for interface parity; tuning hints for NVIDIA parts
        self.k_split_2 = k_split_2
        assert self.in_features % self.group_size == 0
        assert out_features % (32 // self.w_bit) == 0
        pack = 32 // self.w_bit
        self.register_buffer("qweight", torch.zeros((in_features // 4, out_features // pack * 4), dtype=torch.int32, device=dev))
        self.register_buffer("qzeros", torch.zeros((in_features // self.group_size, out_features * 2 // pack), dtype=torch.int32, device=dev))
        self.register_buffer("scales", torch.zeros((in_features // self.group_size, out_features * 2), dtype=torch.float16, device=dev))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features), dtype=torch.float16, device=dev))
        else:
            self.bias = None
        self._layout = _Layout()

    # ---------------------------------------------------------------- layout tracking
    def _key(self):
        # (inference tensors -- anything made under torch.inference_mode(), which is how the reference runs every forward,
        # quick/awq/modules/fused/model.py:76, examples/benchmark.py:45 -- have no version counter)
        return tuple((t.data_ptr(), 0 if t.is_inference() else t._version, t.device) for t in (self.qweight, self.scales, self.qzeros))

    @property
    def is_prepared(self):
        """True when the buffers currently hold MI355X order."""
        lay = self._layout
        if not lay.prepared:
            return False
        if lay.key is None:                 # a copy / unpickled module: same bits in new storages
            lay.key = self._key()
        elif lay.key != self._key():        # rewritten from outside: reference order again
            lay.prepared, lay.key = False, None
        return lay.prepared

    def _set_packed(self, qweight, scales, qzeros, prepared):
        self.qweight, self.scales, self.qzeros = qweight, scales, qzeros
        self._layout = _Layout(prepared, self._key() if prepared else None)

    def prepare(self):
        """Permute reference-order buffers into MI355X order (HIP repack kernels on the GPU)."""
        if self.is_prepared or self.in_features % 128 != 0:   # (K % 128 != 0: stays in reference order, see forward)
            return self
        if self.qweight.is_cuda:
            out = kernels.repack_cuda_to_mi355x(self.qweight.contiguous(), self.scales.contiguous(), self.qzeros.contiguous())
        else:
            out = packing.cuda_to_mi355x(self.qweight, self.scales, self.qzeros)
        self._set_packed(*out, prepared=True)
        return self

    def reference_order(self):
        """(qweight, scales, qzeros) in the reference's packed order, whatever the buffers hold now."""
        if not self.is_prepared:
            return self.qweight, self.scales, self.qzeros
        if self.qweight.is_cuda:
            return kernels.repack_mi355x_to_cuda(self.qweight, self.scales, self.qzeros)
        return packing.mi355x_to_cuda(self.qweight, self.scales, self.qzeros)

    def _apply(self, fn, *args, **kwargs):
        was = self.is_prepared
        super()._apply(fn, *args, **kwargs)
        self._layout = _Layout(was, self._key() if was else None)   # .to()/.cuda() move bits, they do not reorder them
        return self

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # Checkpoints always hold the reference's order, and nn.Module copies them into the existing storages in place.  Do not
        # leave it to the version counters to notice: a deep copy or an unpickled module re-keys itself on first use (a copy of
        # prepared buffers is prepared), and inference tensors have no version counter at all -- so a load that followed either
        # would be taken for MI355X order.  A partial load (some of the three tensors) must not mix the two orders either: a
        # prepared module goes back to the reference's order first.
        if any(prefix + n in state_dict for n in ("qweight", "scales", "qzeros")):
            if self.is_prepared:
                self._set_packed(*self.reference_order(), prepared=False)
            self._layout = _Layout(False)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.is_prepared:
            qw, sc, qz = self.reference_order()
            destination[prefix + "qweight"], destination[prefix + "scales"], destination[prefix + "qzeros"] = qw, sc, qz

    # ---------------------------------------------------------------- construction
    @classmethod
    def from_linear(cls, linear, w_bit, group_size, init_only=False, scales=None, zeros=None, k_split_1=2, k_split_2=8):
        awq_linear = cls(w_bit, group_size, linear.in_features, linear.out_features, linear.bias is not None,
                         linear.weight.device, k_split_1, k_split_2)
        if init_only:  # just prepare for loading sd
            return awq_linear
        assert scales is not None and zeros is not None
        if awq_linear.w_bit != 4:
            raise NotImplementedError("Only 4-bit are supported for now.")
        if linear.bias is not None:
            awq_linear.bias = linear.bias.clone().half()
        G = awq_linear.group_size
        intweight = packing.quantize_intweight(linear.weight.data, scales, zeros, G)          # [K, N]
        s = scales.t().contiguous().half()                                                    # [K/G, N]
        z = zeros.t().contiguous().to(torch.int32)
        if awq_linear.in_features % 128 == 0:
            awq_linear._set_packed(*packing.pack_mi355x(intweight, s, z), prepared=True)
        else:   # not tileable in the MI355X order: the buffers keep the checkpoint order, forward() runs on a padded copy
            awq_linear._set_packed(*packing.pack_cuda_order(intweight, s, z), prepared=False)
        return awq_linear

    # ---------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, x):
        out_shape = x.shape[:-1] + (self.out_features,)
        if self.in_features % 128 != 0:
            # Accepted by the reference (in_features % 32 == 0), not tileable in the MI355X order: the buffers stay in the
            # reference's order and the GEMM runs on a zero-padded MI355X-order copy (kernels.padded_in_features), cached
            # like the function-level drop-in's.
            out = kernels.gemm_forward_cuda_quick(x.reshape(-1, x.shape[-1]), self.qweight, self.scales, self.qzeros, 2)
            if self.bias is not None:
                out = out + self.bias
            return out.reshape(out_shape)
        if not self.is_prepared:
            self.prepare()
        out = kernels.gemm_forward(x.reshape(-1, x.shape[-1]), self.qweight, self.scales, self.qzeros, bias=self.bias)
        return out.reshape(out_shape)

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}, w_bit={}, group_size={}".format(
            self.in_features, self.out_features, self.bias is not None, self.w_bit, self.group_size)
