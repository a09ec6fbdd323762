"""Producing q / k / v on the device (second half).

``Ops::rms_norm`` (``include/ops/normalization.hpp:13-23``) and ``Ops::matmul``
(``include/ops/linear.hpp:12-22``) as ``LlamaModel::forward`` calls them at the head of a layer
(``include/core/model.hpp:161-166``), in the batched form of ``BatchOps``
(``experimental/chunked_prefill/batch_ops.hpp:8-23,58-76``).  Device pointers only: every argument is
anything ``_devptr`` understands (an int address, a ``DeviceArray``, a torch tensor)."""
from __future__ import annotations

from . import _capi
from .kv_pool import _devptr

RMS_EPS = 1e-5  # the reference's default (normalization.hpp:13)


def batch_rms_norm_device(out_dev, in_dev, weight_dev, batch_size: int, dim: int, eps: float = RMS_EPS,
                          device: int = 0, stream=None) -> None:
    """out[b] = in[b] * rms_b * weight (BatchOps::batch_rms_norm)."""
    _capi.check(_capi.lib.nvk_batch_rms_norm_device(_devptr(out_dev), _devptr(in_dev), _devptr(weight_dev),
                                                    int(batch_size), int(dim), float(eps), int(device), _devptr(stream)))


def batch_add_rms_norm_device(out_dev, x_dev, addend_dev, weight_dev, batch_size: int, dim: int, eps: float = RMS_EPS,
                              device: int = 0, stream=None) -> None:
    """x += addend (the residual add, model.hpp:227-228), then out = rms_norm(x), in one kernel."""
    _capi.check(_capi.lib.nvk_batch_add_rms_norm_device(_devptr(out_dev), _devptr(x_dev), _devptr(addend_dev),
                                                        _devptr(weight_dev), int(batch_size), int(dim), float(eps),
                                                        int(device), _devptr(stream)))


def batch_matmul_device(out_dev, in_dev, weight_dev, batch_size: int, in_dim: int, out_dim: int, device: int = 0,
                        stream=None) -> None:
    """out[b][i] = dot(in[b], weight[i]), weight [out_dim, in_dim] (BatchOps::batch_matmul)."""
    _capi.check(_capi.lib.nvk_batch_matmul_device(_devptr(out_dev), _devptr(in_dev), _devptr(weight_dev), int(batch_size),
                                                  int(in_dim), int(out_dim), int(device), _devptr(stream)))


def batch_matmul2_device(out_a_dev, out_b_dev, in_dev, weight_a_dev, weight_b_dev, batch_size: int, in_dim: int,
                          out_dim_a: int, out_dim_b: int, device: int = 0, stream=None) -> None:
    """Two products over the same activations in one launch (w_gate and w_up, model.hpp:232-233)."""
    _capi.check(_capi.lib.nvk_batch_matmul2_device(_devptr(out_a_dev), _devptr(out_b_dev), _devptr(in_dev),
                                                   _devptr(weight_a_dev), _devptr(weight_b_dev), int(batch_size),
                                                   int(in_dim), int(out_dim_a), int(out_dim_b), int(device),
                                                   _devptr(stream)))


def batch_swiglu_ffn_device(hb_dev, hb2_dev, in_dev, w_gate_dev, w_up_dev, batch_size: int, in_dim: int, hidden_dim: int,
                            device: int = 0, stream=None) -> None:
    """hb = silu(w_gate . in) * (w_up . in) (model.hpp:232-234, Ops::swiglu); hb2 is scratch."""
    _capi.check(_capi.lib.nvk_batch_swiglu_ffn_device(_devptr(hb_dev), _devptr(hb2_dev), _devptr(in_dev),
                                                      _devptr(w_gate_dev), _devptr(w_up_dev), int(batch_size),
                                                      int(in_dim), int(hidden_dim), int(device), _devptr(stream)))


def qkv_project_device(q_dev, k_dev, v_dev, x_dev, xb_dev, rms_weight_dev, wq_dev, wk_dev, wv_dev, batch_size: int,
                       dim: int, q_dim: int, kv_dim: int, rms_eps: float = RMS_EPS, device: int = 0, stream=None) -> None:
    """xb = rms_norm(x, rms_weight); q, k, v = wq.xb, wk.xb, wv.xb in one launch (model.hpp:161-166)."""
    _capi.check(_capi.lib.nvk_qkv_project_device(_devptr(q_dev), _devptr(k_dev), _devptr(v_dev), _devptr(x_dev),
                                                 _devptr(xb_dev), _devptr(rms_weight_dev), _devptr(wq_dev),
                                                 _devptr(wk_dev), _devptr(wv_dev), int(batch_size), int(dim), int(q_dim),
                                                 int(kv_dim), float(rms_eps), int(device), _devptr(stream)))
