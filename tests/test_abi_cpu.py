"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares, validates
descriptors and reports errors through the status / gc_last_error convention.  No kernel is launched."""
import ctypes
import os
import re

import pytest
import torch

from gan_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ganconv.h")).read()
    declared = set(re.findall(r"\b(gc_[a-z0-9_]+)\s*\(", header))
    declared -= {"gc_pack_weight(desc"}  # nothing: keeps the set literal explicit
    assert set(_lib.EXPORTS) == declared, (sorted(declared), sorted(_lib.EXPORTS))
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.gc_version() == 1


def _desc(**kw):
    d = _lib.ConvDesc()
    base = dict(N=2, H=16, W=16, Cin=8, Ho=8, Wo=8, Cout=16, kh=4, kw=4, stride_h=2, stride_w=2, pad_h=1, pad_w=1,
                x_dtype=_lib.GC_F32, y_dtype=_lib.GC_F32, precision=_lib.GC_PREC_BF16)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    return d


def test_sizes_and_descriptor_validation():
    L = _lib.lib()
    d = _desc()
    # forward image, thin input (Cin <= 32): [Cout][16 taps x 8 channels] bf16; wide input: [Cout][16 taps][Cin padded to 64]
    assert L.gc_packed_weight_bytes(ctypes.byref(d), _lib.GC_PASS_FPROP) == 16 * 16 * 8 * 2
    assert L.gc_packed_weight_bytes(ctypes.byref(_desc(Cin=72)), _lib.GC_PASS_FPROP) == 16 * 16 * 128 * 2
    # data-gradient image of a stride-2 conv: [Cin][16 taps][Cout padded to 64]
    assert L.gc_packed_weight_bytes(ctypes.byref(d), _lib.GC_PASS_DGRAD) == 8 * 16 * 64 * 2
    d.precision = _lib.GC_PREC_F32X3
    assert L.gc_packed_weight_bytes(ctypes.byref(d), _lib.GC_PASS_FPROP) == 2 * 16 * 16 * 8 * 2
    # per split: one fp32 slab of the weight's (padded) shape and one row [Cout] for the bias partial sums
    need = L.gc_wgrad_workspace_bytes(ctypes.byref(d))
    assert need > 0 and need % (16 * 16 * 8 * 4 + 16 * 4) == 0
    assert L.gc_workspace_bytes(ctypes.byref(d), _lib.GC_PASS_WGRAD) == need
    assert L.gc_workspace_bytes(ctypes.byref(d), _lib.GC_PASS_FPROP) == d.N * d.Ho * d.Wo * d.Cout * 128
    bad = _desc(Ho=9)
    assert L.gc_packed_weight_bytes(ctypes.byref(bad), _lib.GC_PASS_FPROP) == 0
    assert b"do not match" in L.gc_last_error()
    assert L.gc_conv2d_fprop(ctypes.byref(bad), None, None, None, None, None) == -1          # GC_BAD_DESC
    assert L.gc_conv2d_fprop(ctypes.byref(_desc(stride_h=3, stride_w=3, Ho=5, Wo=5)), None, None, None, None, None) == -2
    assert L.gc_conv2d_fprop(ctypes.byref(_desc()), None, None, None, None, None) == -5       # null pointers
    assert L.gc_conv2d_fprop(ctypes.byref(_desc(N=0)), None, None, None, None, None) == 0     # empty batch is a no-op
    assert L.gc_conv2d_wgrad(ctypes.byref(_desc()), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16),
                             (ctypes.c_int64 * 4)(1, 1, 1, 1), 1.0, 0, None, None, 0, None) == -3  # workspace too small
    assert L.gc_dwblur3x3(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 4, 4, 6, _lib.GC_F32, None) == -2
    # the entry point with the built-in scratch rejects problems whose partial sums exceed it, before touching a GPU
    big = _desc(N=64, H=64, W=64, Cin=1024, Ho=32, Wo=32, Cout=1024)
    assert L.gc_wgrad_workspace_bytes(ctypes.byref(big)) > (32 << 20)
    assert L.gc_conv2d_wgrad_noalloc(ctypes.byref(big), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16),
                                     (ctypes.c_int64 * 4)(1, 1, 1, 1), 1.0, 0, None, None) == -3
    assert b"built-in scratch" in L.gc_last_error()


def test_argument_validation_of_the_layer_kernels():
    """Entry points around the convolutions reject bad shapes, null or misaligned pointers before touching the GPU."""
    L = _lib.lib()
    q = ctypes.c_void_p(256)      # an aligned non-null pointer that is never dereferenced: every call below fails validation
    odd = ctypes.c_void_p(260)
    # attention: N a multiple of 128, d in {8, 16}, C in {64, 128}
    assert L.gc_attention_fwd(q, q, q, q, q, 2, 100, 8, 64, None) == -2 and b"N % 128" in L.gc_last_error()
    assert L.gc_attention_fwd(q, q, q, q, q, 2, 256, 12, 64, None) == -2
    assert L.gc_attention_fwd(q, q, q, q, q, 2, 256, 8, 96, None) == -2
    assert L.gc_attention_fwd(q, q, None, q, q, 2, 256, 8, 64, None) == -5
    assert L.gc_attention_fwd(q, odd, q, q, q, 2, 256, 8, 64, None) == -5
    assert L.gc_attention_bwd(q, q, q, q, q, q, q, q, q, None, 2, 256, 8, 64, None) == -5
    # pooling / replication: channels a multiple of 8
    assert L.gc_pool2x2(q, q, 1, 4, 4, 12, _lib.GC_BF16, 0.25, None) == -2
    assert L.gc_replicate2x(q, None, 1, 4, 4, 16, _lib.GC_BF16, 1.0, None) == -5
    assert L.gc_pool2x2(q, q, 0, 4, 4, 16, _lib.GC_F32, 0.25, None) == 0          # empty batch
    # conditional batchnorm: workspace size grows with the sample-aligned slabs; gain and bias are mandatory
    ws = L.gc_cond_batchnorm_workspace_bytes(4, 1024, 64)
    assert ws >= (4 * 64 * 2 + 2 * 64) * 4 and L.gc_cond_batchnorm_workspace_bytes(0, 1024, 64) == 0
    assert L.gc_cond_batchnorm_act_fwd(q, q, None, None, None, None, None, q, q, q, q, ws, 4, 1024, 64, _lib.GC_F32, 1e-5, 0.1, 1, 0.0, 1, None) == -5
    assert L.gc_cond_batchnorm_act_fwd(q, q, q, q, None, None, None, q, q, q, q, 16, 4, 1024, 64, _lib.GC_F32, 1e-5, 0.1, 1, 0.0, 1, None) == -3
    assert L.gc_cond_batchnorm_act_fwd(q, q, q, q, None, None, None, q, q, q, q, ws, 4, 1024, 66, _lib.GC_F32, 1e-5, 0.1, 1, 0.0, 1, None) == -2
    # noise -> LeakyReLU -> InstanceNorm -> style chain
    assert L.gc_noise_lrelu_adain_fwd(q, q, q, q, q, q, q, q, None, 0, 2, 64, 30, _lib.GC_BF16, 1e-5, 0.2, None) == -2
    assert L.gc_noise_lrelu_adain_fwd(q, q, q, q, q, q, q, q, None, 0, 2, 64, 32, _lib.GC_BF16, 1e-5, 0.0, None) == -2     # slope must be positive
    assert L.gc_noise_lrelu_adain_fwd(q, None, q, q, q, q, q, q, None, 0, 2, 64, 32, _lib.GC_BF16, 1e-5, 0.2, None) == -5
    assert L.gc_noise_lrelu_adain_fwd(q, q, q, q, q, q, q, q, None, 0, 0, 64, 32, _lib.GC_BF16, 1e-5, 0.2, None) == 0
    assert L.gc_noise_lrelu_adain_workspace_bytes(4, 64, 64) == 0 and L.gc_noise_lrelu_adain_workspace_bytes(4, 65536, 64) >= 4 * 2 * 64 * 12
    assert L.gc_noise_lrelu_adain_fwd(q, q, q, q, q, q, q, q, None, 0, 4, 65536, 64, _lib.GC_BF16, 1e-5, 0.2, None) == -3    # large image: workspace
    # the split-reduction switch reports the previous setting
    assert L.gc_set_split_reduction(0) == 1 and L.gc_set_split_reduction(1) == 0 and L.gc_set_split_reduction(1) == 1


def test_python_layer_fails_loudly_without_gpu():
    import gan_amd
    conv = gan_amd.nn.Conv2d(3, 8, 4, 2, 1)
    assert isinstance(conv, torch.nn.Conv2d) and conv.weight.shape == (8, 3, 4, 4)
    with pytest.raises(RuntimeError, match="GPU"):
        conv(torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="dilation"):
        gan_amd.functional.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(8, 3, 3, 3), dilation=2)
    with pytest.raises(RuntimeError, match="groups"):
        gan_amd.functional.conv2d(torch.zeros(1, 4, 8, 8), torch.zeros(4, 1, 3, 3), groups=4)
    convt = gan_amd.nn.ConvTranspose2d(8, 3, 4, 2, 1, bias=False)
    assert isinstance(convt, torch.nn.ConvTranspose2d) and convt.weight.shape == (8, 3, 4, 4)


def test_state_dict_names_match_reference_checkpoints():
    from gan_amd.harness.basic_gans import Discriminator, Generator
    from oracle import ref_models as R
    G, D = Generator(100, 8, 3, 64), Discriminator(3, 8, 64, is_wgan=True)
    Gr, Dr = R.Generator(100, 8, 3, 64), R.Discriminator(3, 8, 64, is_wgan=True)
    assert list(G.state_dict().keys()) == list(Gr.state_dict().keys())
    assert list(D.state_dict().keys()) == list(Dr.state_dict().keys())
    assert [tuple(v.shape) for v in G.state_dict().values()] == [tuple(v.shape) for v in Gr.state_dict().values()]
    G.load_state_dict(Gr.state_dict()); D.load_state_dict(Dr.state_dict())
