"""Host-side calls of the C ABI that need no device: argument validation, workspace queries, and launch sequences that
stop at the first HIP error on a box without a GPU.  No torch import: tests/test_host_cpu.py runs them against the shipped
library, tests/san_driver.py against the AddressSanitizer / UndefinedBehaviorSanitizer build of the same host code."""
import ctypes


def round2_cases(L, _lib):
    arr = (ctypes.c_int * 2)(128, 64)
    assert L.gfs3d_pointwise_workspace_bytes_ex(2, 256, 1, 256, 2, arr) > 0
    assert L.gfs3d_pointwise_workspace_bytes_ex(2, 256, 1, 100, 2, arr) == 0          # segment width not a multiple of 64
    assert L.gfs3d_pointwise_workspace_bytes_ex(2, 256, 5, 64, 2, arr) == 0           # too many segments
    assert L.gfs3d_linear_workspace_bytes(2, 256, 256, 192) > 0
    assert L.gfs3d_linear_workspace_bytes(2, 256, 100, 192) == 0 and L.gfs3d_linear_workspace_bytes(2, 256, 256, 320) == 0
    a = _lib.PackArgs()
    assert L.gfs3d_pack_rows_f32(None, None) != 0 and b"NULL args" in L.gfs3d_last_error()
    a.B, a.N, a.n_src = 1, 64, 0
    assert L.gfs3d_pack_rows_f32(ctypes.byref(a), None) != 0 and b"n_src" in L.gfs3d_last_error()
    a.n_src, a.dst, a.dst_ld = 1, 1, 16
    a.src[0], a.width[0] = 1, 64
    assert L.gfs3d_unpack_rows_f32(ctypes.byref(a), None) != 0 and b"dst_ld" in L.gfs3d_last_error()
    ptrs = (ctypes.c_void_p * 3)(None, None, None)
    assert L.gfs3d_linear_forward_f32(None, 1, 64, 256, None, 192, ptrs, 0, None, 0, None) != 0
    assert L.gfs3d_linear_forward_f32(1, 1, 64, 100, 1, 192, ptrs, 0, 1, 1 << 30, None) != 0 and b"K=100" in L.gfs3d_last_error()
    assert L.gfs3d_attention_forward_mixed(1, 1, 1, 1, 64, 64, 0.125, 0.0, 0, None, 1, 1, 7, None) != 0 \
        and b"compute" in L.gfs3d_last_error()
    assert L.gfs3d_attention_forward_mixed(1, 1, 1, 1, 32, 64, 0.125, 0.0, 0, None, 1, 1, 0, None) != 0 \
        and b"head dimension" in L.gfs3d_last_error()


def round3_cases(L, _lib):
    assert L.gfs3d_capl_head_workspace_bytes(32, 2048, 13) > 0 and L.gfs3d_capl_head_workspace_bytes(64, 2048, 21) > 0
    assert L.gfs3d_capl_head_workspace_bytes(32, 2048, 24) == 0 and L.gfs3d_capl_head_workspace_bytes(0, 2048, 13) == 0
    assert L.gfs3d_capl_head_forward_phase(None, 0) != 0 and b"args is NULL" in L.gfs3d_last_error()
    a = _lib.CaplHeadArgs()
    a.B, a.C, a.N, a.classes, a.base_num = 4, 64, 256, 13, 7
    assert L.gfs3d_capl_head_forward_phase(ctypes.byref(a), 0) != 0 and b"feature channels" in L.gfs3d_last_error()
    a.C, a.n_fake = 128, 20
    assert L.gfs3d_capl_head_forward_phase(ctypes.byref(a), 0) != 0 and b"fake-novel" in L.gfs3d_last_error()
    a.n_fake, a.half, a.feat_bstride = 2, 2, 128 * 256
    assert L.gfs3d_capl_head_forward_phase(ctypes.byref(a), 0) != 0 and b"NULL buffer" in L.gfs3d_last_error()
    for name in ("feat", "label", "main_proto", "bg_proto", "fsum", "workspace", "loss"):
        setattr(a, name, 1)                             # (non-NULL placeholders: the checks below stop before any use)
    a.workspace_bytes = 16
    assert L.gfs3d_capl_head_forward_phase(ctypes.byref(a), 0) != 0 and b"workspace too small" in L.gfs3d_last_error()
    a.workspace_bytes = 1 << 30
    a.fake[0], a.fake[1] = 3, 99
    assert L.gfs3d_capl_head_backward_phase(ctypes.byref(a), 1) != 0 and b"fake-novel label 99" in L.gfs3d_last_error()
    a.fake[1] = 5
    assert L.gfs3d_capl_head_forward_phase(ctypes.byref(a), 7) != 0 and b"phase" in L.gfs3d_last_error()
    assert L.gfs3d_capl_head_backward_phase(ctypes.byref(a), 0) != 0 and b"NULL gradient buffer" in L.gfs3d_last_error()
    e = _lib.CaplEvalArgs()
    e.B, e.C, e.N, e.classes, e.base_num, e.Kw = 2, 128, 64, 13, 7, 4000
    assert L.gfs3d_capl_head_eval(ctypes.byref(e)) != 0 and b"coding" in L.gfs3d_last_error()
    assert L.gfs3d_label_presence(None, 10, None, None) != 0 and b"bad argument" in L.gfs3d_last_error()
    assert L.gfs3d_kmeans_assign_fast_f32(1, 100, 64, 1, 40, 1, 1, 1, None, None, 0, None) != 0 and b"workspace" in L.gfs3d_last_error()


def round5_cases(L, _lib):
    """The seeding entry points, and the EdgeConv forward / backward argument checks down to the workspace plan."""
    assert L.gfs3d_kmeanspp_greedy_workspace_bytes(2100000, 192, 200) > L.gfs3d_kmeanspp_workspace_bytes(2100000, 200) > 0
    assert L.gfs3d_kmeanspp_greedy_workspace_bytes(0, 192, 200) == 0
    assert L.gfs3d_kmeanspp_greedy_f32(None, 10, 8, 2, 0, None, 2, None, None, None, 0, 1, None) != 0 and b"NULL" in L.gfs3d_last_error()
    assert L.gfs3d_kmeanspp_greedy_f32(1, 10, 8, 20, 0, 1, 2, 1, 1, 1, 1 << 30, 1, None) != 0 and b"K <= M" in L.gfs3d_last_error()
    assert L.gfs3d_kmeanspp_greedy_f32(1, 10, 300, 2, 0, 1, 2, 1, 1, 1, 1 << 30, 1, None) != 0 and b"C=300" in L.gfs3d_last_error()
    assert L.gfs3d_kmeanspp_greedy_f32(1, 10, 8, 2, 0, 1, 9, 1, 1, 1, 1 << 30, 1, None) != 0 and b"n_trials" in L.gfs3d_last_error()
    assert L.gfs3d_kmeanspp_greedy_f32(1, 10, 8, 2, 11, 1, 2, 1, 1, 1, 1 << 30, 1, None) != 0 and b"first" in L.gfs3d_last_error()
    assert L.gfs3d_kmeanspp_greedy_f32(1, 10, 8, 2, 0, 1, 2, 1, 1, 1, 16, 1, None) != 0 and b"workspace" in L.gfs3d_last_error()
    assert L.gfs3d_kmeanspp_seed_f32(1, 10, 8, 2, 0, 1, 1, 1, 1, 16, 1, None) != 0 and b"workspace" in L.gfs3d_last_error()
    assert L.gfs3d_kmeans_fix_step(1, 1, 200, 192, 30, 1, None, 1, 1, 1, 1, None) != 0 and b"NULL" in L.gfs3d_last_error()
    assert L.gfs3d_kmeans_fix_step(1, 1, 200, 192, 61, 1, 1, 1, 1, 1, 1, None) != 0 and b"shift=61" in L.gfs3d_last_error()
    assert L.gfs3d_kmeans_fix_step(1, 1, 1 << 12, 1 << 12, 30, 1, 1, 1, 1, 1, 1, None) != 0 and b"bad argument" in L.gfs3d_last_error()
    a = _lib.EdgeConvArgs()
    a.B, a.C, a.N, a.k, a.training = 2, 9, 256, 20, 1
    assert L.gfs3d_edgeconv_forward_f32(ctypes.byref(a), None) != 0 and b"NULL input" in L.gfs3d_last_error()
    for name in ("x", "idx", "w1", "bn1_w", "bn1_b", "w2", "bn2_w", "bn2_b", "P", "Q", "stats", "workspace"):
        setattr(a, name, 1)
    a.workspace_bytes = 64
    assert L.gfs3d_edgeconv_forward_f32(ctypes.byref(a), None) != 0 and b"workspace" in L.gfs3d_last_error()
    a.workspace_bytes = L.gfs3d_edgeconv_workspace_bytes(2, 9, 256, 20)
    assert a.workspace_bytes > 0
    assert L.gfs3d_edgeconv_forward_f32(ctypes.byref(a), None) != 0 and b"no output buffer" in L.gfs3d_last_error()
    a.y_pm = 1
    assert L.gfs3d_edgeconv_forward_f32(ctypes.byref(a), None) != 0 and b"training mode needs" in L.gfs3d_last_error()
    a.k = 300
    assert L.gfs3d_edgeconv_forward_f32(ctypes.byref(a), None) != 0 and b"k" in L.gfs3d_last_error()
    b = _lib.EdgeConvBwdArgs()
    assert L.gfs3d_edgeconv_backward_f32(None, None) != 0 and b"args is NULL" in L.gfs3d_last_error()
    assert L.gfs3d_edgeconv_backward_f32(ctypes.byref(b), None) != 0


def round6_cases(L, _lib):
    """Round 6: the softmax backward from packed rows, the multi-tensor Adam, the storage-format fields of the point-wise stack."""
    f = L.gfs3d_word_soft_backward_rows
    assert f(None, 0, 384, 1, 2, 180, 64, 1, None) != 0 and b"NULL" in L.gfs3d_last_error()
    assert f(16, 3, 384, 16, 2, 180, 64, 16, None) != 0 and b"dtype" in L.gfs3d_last_error()
    assert f(16, 0, 100, 16, 2, 180, 64, 16, None) != 0 and b"bad shape" in L.gfs3d_last_error()
    assert f(16, 0, 388, 16, 2, 180, 64, 16, None) != 0 and b"bad shape" in L.gfs3d_last_error()          # ld % 8
    assert f(16, 0, 384, 16, 2, 300, 64, 16, None) != 0 and b"bad shape" in L.gfs3d_last_error()          # Kw > 256
    assert f(20, 1, 384, 16, 2, 180, 64, 16, None) != 0 and b"aligned" in L.gfs3d_last_error()
    a = _lib.AdamArgs()
    assert L.gfs3d_adam_multi_f32(None, None) != 0 and b"NULL args" in L.gfs3d_last_error()
    assert L.gfs3d_adam_multi_f32(ctypes.byref(a), None) != 0 and b"n_tensors" in L.gfs3d_last_error()
    a.n_tensors = 1
    assert L.gfs3d_adam_multi_f32(ctypes.byref(a), None) != 0 and b"bias corrections" in L.gfs3d_last_error()
    a.bias_correction1 = a.bias_correction2 = 0.5
    assert L.gfs3d_adam_multi_f32(ctypes.byref(a), None) != 0 and b"NULL or empty tensor 0" in L.gfs3d_last_error()
    a.n_tensors = 65
    assert L.gfs3d_adam_multi_f32(ctypes.byref(a), None) != 0 and b"n_tensors" in L.gfs3d_last_error()
    # x_type / dx_type need 16-bit storage in a reduced training call
    pa = _lib.PointwiseArgs()
    pa.B, pa.N, pa.n_seg, pa.n_layers, pa.training = 2, 64, 1, 1, 1
    pa.width[0] = 64
    for name in ("x_seg", "w", "bn_w", "bn_b", "h", "stats"):
        getattr(pa, name)[0] = 16
    pa.y_cf, pa.workspace, pa.workspace_bytes = 16, 16, 1 << 30
    pa.x_type = _lib.COMPUTE_F16
    assert L.gfs3d_pointwise_forward_f32(ctypes.byref(pa), None) != 0 and b"x_type" in L.gfs3d_last_error()
    pk = _lib.PackArgs()
    pk.B, pk.N, pk.n_src, pk.dst, pk.dst_ld, pk.dst_type = 1, 64, 1, 16, 64, 7
    pk.src[0], pk.width[0] = 16, 64
    assert L.gfs3d_pack_rows_f32(ctypes.byref(pk), None) != 0 and b"dst_type" in L.gfs3d_last_error()
    pk.dst_type = _lib.COMPUTE_F16
    assert L.gfs3d_unpack_rows_f32(ctypes.byref(pk), None) != 0 and b"dst_type" in L.gfs3d_last_error()        # half: packing only


def workspace_sweep(L):
    """Every workspace query over a grid of shapes, odd ones included: the plans are pure host arithmetic (sizes, alignments,
    grid dimensions) - monotone in the batch and never absurd."""
    for C in (3, 9, 64, 128):
        prev = 0
        for B in (1, 2, 7, 32, 64):
            for N in (20, 33, 256, 2048, 4099):
                w = L.gfs3d_edgeconv_workspace_bytes(B, C, N, 20)
                assert 0 < w < (1 << 40)
            w = L.gfs3d_edgeconv_workspace_bytes(B, C, 2048, 20)
            assert w >= prev
            prev = w
    assert L.gfs3d_edgeconv_workspace_bytes(0, 9, 2048, 20) == 0
    for M in (1, 1023, 1024, 1025, 70000, 2100000):
        for K in (1, 13, 200, 256):
            assert L.gfs3d_kmeans_workspace_bytes(M, 192, K) >= 0
            assert L.gfs3d_kmeanspp_workspace_bytes(M, K) > 0
            assert L.gfs3d_kmeanspp_greedy_workspace_bytes(M, 192, K) > 0
    for K in (1, 13, 33, 129, 180, 200, 256):
        for C in (64, 128, 192):
            assert L.gfs3d_proto_backward_feat_workspace_bytes(C, K) >= 0
            assert L.gfs3d_proto_backward_proto_workspace_bytes(32, 2048, K) >= 0
    for classes in (2, 13, 14, 21, 22):
        for B in (1, 32, 64):
            assert L.gfs3d_capl_head_workspace_bytes(B, 2048, classes) > 0
