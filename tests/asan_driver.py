"""Run under LD_PRELOAD=<libclang_rt.asan> by tests/test_abi_cpu.py (CPU box only): drives the host side of every
C-ABI family of the sanitizer build (csrc/build_asan.sh: AddressSanitizer + UBSan, no device code) with null pointers,
bad shapes, short workspaces, misaligned operands and plausible full-size geometries.  With fake non-null device
pointers an entry point runs ALL its host code (tile / split selection, workspace carving, grid arithmetic) up to the
launch, which fails cleanly without a GPU.  Any sanitizer report aborts the process; the test asserts exit code 0."""
import ctypes
import sys

sys.path.insert(0, sys.argv[2])
from weatherforecastingtoolkit_amd import _lib  # noqa: E402

lib = _lib.load(sys.argv[1])
P = 0x7F0000000000          # fake, 16-byte aligned "device" addresses (never dereferenced on the host)
Q, R, S = P + (1 << 32), P + (2 << 32), P + (3 << 32)
WS = P + (8 << 32)
checked = 0


def expect(rc, want, what):
    global checked
    checked += 1
    ok = rc in want if isinstance(want, (tuple, list, set)) else rc == want
    if not ok:
        print(f"FAIL {what}: rc={rc} want={want} msg={lib.wfae_last_error_string()}")
        sys.exit(3)


NULL, SHAPE, WSP = -2, -1, None
codes = {}
for line in open(_lib.HEADER):
    line = line.strip()
    if line.startswith("#define WFAE_ERR_") or line.startswith("WFAE_ERR_"):
        parts = line.replace("=", " ").replace(",", " ").split()
        try:
            codes[[p for p in parts if p.startswith("WFAE_ERR_")][0]] = int(parts[-1])
        except (ValueError, IndexError):
            pass
NULL = codes.get("WFAE_ERR_NULL_POINTER", -2)
SHAPE = codes.get("WFAE_ERR_BAD_SHAPE", -1)
WSP = codes.get("WFAE_ERR_WORKSPACE", -4)
LAUNCH = codes.get("WFAE_ERR_LAUNCH", -5)
ANY_FAIL = set(range(-16, 0))

# --- null pointers / bad shapes are rejected before anything else
expect(lib.wfae_conv1x1_fwd(None, None, None, None, 0, None, 1, 1, 1, 1, None), NULL, "conv1x1_fwd null")
expect(lib.wfae_conv1x1_fwd(P, Q, None, None, 0, R, 0, 1, 1, 1, None), SHAPE, "conv1x1_fwd NB=0")
expect(lib.wfae_conv1x1_fwd(P, Q, None, None, 0, R, 1 << 16, 8, 8, 1 << 16, None), SHAPE, "conv1x1_fwd NB*HW overflow")
expect(lib.wfae_adamw(None, None, None, None, 10, 0.1, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.001, 1.0, None), NULL, "adamw null")
expect(lib.wfae_conv1x1_bwd_weight(P, Q, R, 32, 128, 32, 384 * 384, 0, WS, 16, None), WSP, "conv1x1 wgrad short workspace")
expect(lib.wfae_linear_fwd(P, Q, None, R, 32, 36864, 2048, WS, 1024, None), WSP, "linear_fwd short workspace")
# --- transformer entry points: alignment of the float4-read qkv buffer, dropout range in fwd AND bwd
expect(lib.wfae_mha_fwd(P + 4, Q, R, 32, 64, 8, 8, 0, 0.0, 1, None), SHAPE, "mha_fwd misaligned qkv")
expect(lib.wfae_mha_fwd(P, Q, R, 32, 64, 8, 8, 0, 1.0, 1, None), SHAPE, "mha_fwd p_drop = 1")
expect(lib.wfae_mha_bwd(P, Q, R, S, 32, 64, 8, 8, 0, 1.0, 1, None), SHAPE, "mha_bwd p_drop = 1")
expect(lib.wfae_mha_bwd(P, Q, R, S, 32, 64, 8, 8, 0, -0.5, 1, None), SHAPE, "mha_bwd p_drop < 0")
expect(lib.wfae_mha_fwd(P, Q, R, 65, 64, 8, 8, 0, 0.0, 1, None), SHAPE, "mha_fwd S > 64")

# --- full host paths at the model's real geometries (B = 32, 384x384): tile / split-K selection, workspace carving and
#     grid arithmetic all run; the launch itself fails without a GPU (any negative status is fine, a sanitizer report
#     or a crash is not)
big = 1 << 33
for (c, h) in [(256, 192), (512, 96), (1024, 48), (1024, 24), (128, 384)]:
    hw, mid = h * h, c // 4
    expect(lib.wfae_conv1x1_fwd(P, Q, None, None, 0, R, 32, c, mid, hw, None), ANY_FAIL, f"conv1x1_fwd {c}@{h}")
    expect(lib.wfae_conv1x1_fwd(P, Q, None, S, c * hw, R, 32, mid, c, hw, None), ANY_FAIL, f"conv1x1_fwd+res {c}@{h}")
    expect(lib.wfae_conv1x1_bwd_data(P, Q, R, 32, c, mid, hw, None), ANY_FAIL, f"conv1x1_bwd_data {c}@{h}")
    expect(lib.wfae_conv1x1_bwd_weight(P, Q, R, 32, c, mid, hw, 0, WS, big, None), ANY_FAIL, f"conv1x1_bwd_weight {c}@{h}")
    expect(lib.wfae_gconv3x3_fwd(P, Q, R, 32, mid, h, h, 8, 0, WS, big, None), ANY_FAIL, f"gconv3x3_fwd {mid}@{h}")
    expect(lib.wfae_gconv3x3_bwd_weight(P, Q, R, 32, mid, h, h, 8, 0, WS, big, None), ANY_FAIL, f"gconv3x3_wgrad {mid}@{h}")
    expect(lib.wfae_bn_stats_train(P, 32, c, hw, Q, R, 1e-5, 0.1, S, S + 4096, S + 8192, S + 12288, S + 16384, S + 20480,
                                   WS, big, None), ANY_FAIL, f"bn_stats_train {c}@{h}")
out = (ctypes.c_int64 * 4)()
for variant in (0, 1):
    for key in [(32, 256, 512, 96, 96), (32, 1024, 1024, 24, 24), (32, 128, 256, 192, 192), (1, 16, 16, 2, 2), (2, 32, 16, 6, 10)]:
        rc = lib.wfae_wino_sizes(variant, *key, ctypes.cast(out, ctypes.c_void_p))
        checked += 1
        if rc == 0:
            assert all(int(v) > 0 for v in out), (variant, key, list(out))
expect(lib.wfae_wino_sizes(1, 32, 256, 512, 0, 96, ctypes.cast(out, ctypes.c_void_p)), ANY_FAIL, "wino_sizes Hlo=0")
# --- split-operand GEMMs: argument validation, the support query and the full host paths (tile / K-split / workspace carving)
UNS = codes.get("WFAE_ERR_UNSUPPORTED", -3)
expect(lib.wfae_split_gemm(0, 3, None, Q, R, 64, 64, 64, 1, None), NULL, "split_gemm null")
expect(lib.wfae_split_gemm(2, 3, P, Q, R, 64, 64, 64, 1, None), SHAPE, "split_gemm bad kind")
expect(lib.wfae_split_gemm(0, 3, P, Q, R, 64, 64, 48, 1, None), UNS, "split_gemm K % 32")
expect(lib.wfae_split_gemm(0, 3, P + 2, Q, R, 64, 64, 64, 1, None), SHAPE, "split_gemm misaligned planes")
expect(lib.wfae_split_gemm(0, 2, P, Q, R, 64, 64, 64, 1, None), SHAPE, "split_gemm planes = 2")
expect(lib.wfae_split_gemm(0, 3, P, Q, R, 65536, 64, 32768, 1, None), SHAPE, "split_gemm operand plane >= 2^31 elements (32-bit loader offsets)")
expect(lib.wfae_split_gemm(1, 3, P, Q, R, 64, 65536, 32768, 1, None), SHAPE, "split_gemm B plane >= 2^31 elements")
expect(lib.wfae_split_bf16x3(P, None, 16, 3, None), NULL, "split_bf16x3 null")
expect(lib.wfae_split_bf16x3(P, Q, 0, 3, None), SHAPE, "split_bf16x3 n = 0")
assert lib.wfae_wino_split_supported(1, 32, 256, 512, 96, 96) == 1 and lib.wfae_wino_split_supported(1, 1, 256, 512, 8, 8) == 0
assert lib.wfae_wino_split_supported(1, 32, 12, 512, 96, 96) == 0 and lib.wfae_wino_split_supported(7, 32, 256, 512, 96, 96) == 0
checked += 2
expect(lib.wfae_wino_gemm_down_split(1, P, Q, R, 3, 1, 256, 512, 8, 8, None), UNS, "wino_gemm_down_split unsupported geometry")
expect(lib.wfae_wino_gemm_wgrad_split(1, P, Q, R, 3, 32, 256, 512, 96, 96, 0, WS, 16, None), WSP, "wino_gemm_wgrad_split short workspace")
for planes in (3, 1):
  for key in [(32, 256, 512, 96, 96), (32, 1024, 1024, 24, 24), (32, 128, 256, 192, 192)]:
    expect(lib.wfae_wino_weights_split(1, P, Q, R, planes, key[1], key[2], None), ANY_FAIL, f"wino_weights_split {key}")
    expect(lib.wfae_wino_in_split(1, P, Q, planes, key[0], key[1], key[3], key[4], None), ANY_FAIL, f"wino_in_split {key}")
    expect(lib.wfae_wino_out_t_split(1, P, Q, planes, key[0], key[2], key[3], key[4], None), ANY_FAIL, f"wino_out_t_split {key}")
    expect(lib.wfae_wino_gemm_down_split(1, P, Q, R, planes, *key, None), ANY_FAIL, f"wino_gemm_down_split {key}")
    expect(lib.wfae_wino_gemm_up_split(1, P, Q, R, planes, *key, None), ANY_FAIL, f"wino_gemm_up_split {key}")
    expect(lib.wfae_wino_gemm_wgrad_split(1, P, Q, R, planes, *key, 0, WS, big, None), ANY_FAIL, f"wino_gemm_wgrad_split {key}")
expect(lib.wfae_wino_gemm_up_split(1, P, Q, R, 2, 32, 256, 512, 96, 96, None), SHAPE, "wino_gemm_up_split planes = 2")
assert lib.wfae_set_split_gemm(0) == 0 and lib.wfae_get_split_gemm() == 0 and lib.wfae_set_split_gemm(1) == 0 and lib.wfae_get_split_gemm() == 1
checked += 1
# grouped 3x3 on the matrix pipe (csrc/g3b.hip): shape table, precision mode, workspace, alignment; every model geometry up to the launch
assert lib.wfae_g3b_supported(32, 384, 384, 8) == 1 and lib.wfae_g3b_supported(64, 192, 192, 8) == 1
assert lib.wfae_g3b_supported(256, 24, 24, 8) == 1 and lib.wfae_g3b_supported(256, 48, 48, 8) == 1
assert lib.wfae_g3b_supported(32, 20, 36, 8) == 0 and lib.wfae_g3b_supported(64, 101, 192, 8) == 0 and lib.wfae_g3b_supported(64, 192, 384, 8) == 0
assert lib.wfae_g3b_f32_supported(128, 96, 96, 8, 1) == 1 and lib.wfae_g3b_f32_supported(64, 192, 192, 8, 1) == 0
assert lib.wfae_g3b_f32_supported(128, 96, 96, 8, 0) == 1 and lib.wfae_g3b_f32_supported(64, 192, 192, 8, 0) == 0
checked += 5
expect(lib.wfae_g3b_fwd_bf16(None, Q, R, 32, 32, 384, 384, 8, 0, WS, big, None), NULL, "g3b_fwd null")
expect(lib.wfae_g3b_fwd_bf16(P, Q, R, 32, 32, 20, 36, 8, 0, WS, big, None), UNS, "g3b_fwd unsupported shape")
expect(lib.wfae_g3b_fwd_bf16(P, Q, R, 32, 32, 384, 384, 8, 0, WS, big, None), UNS, "g3b_fwd needs bf16 precision")
expect(lib.wfae_g3b_bwd_weight_bf16(P, Q, R, 32, 32, 384, 384, 8, 0, WS, big, None), UNS, "g3b_bwd_weight needs bf16 precision")
expect(lib.wfae_g3b_fwd(P, Q, R, 32, 32, 384, 384, 8, 0, WS, big, None), UNS, "g3b_fwd fp32: 4 per group not served")
expect(lib.wfae_g3b_bwd_weight(P, Q, R, 32, 64, 192, 192, 8, 0, WS, big, None), UNS, "g3b_bwd_weight fp32: W = 192 not served")
expect(lib.wfae_g3b_bwd_weight(P, Q, R, 32, 128, 96, 96, 8, 0, WS, 1024, None), WSP, "g3b_bwd_weight fp32 short workspace")
for c, h in [(128, 96), (256, 48), (256, 24)]:
    expect(lib.wfae_g3b_bwd_weight(P, Q, R, 32, c, h, h, 8, 0, WS, big, None), ANY_FAIL, f"g3b_bwd_weight fp32 {c}@{h}")
expect(lib.wfae_g3b_fwd(P, Q, R, 32, 128, 96, 96, 8, 1, WS, big, None), ANY_FAIL, "g3b_fwd fp32 128@96")
expect(lib.wfae_g3b_fwd(P, Q, R, 32, 64, 192, 192, 8, 0, WS, big, None), UNS, "g3b_fwd fp32 64@192: 8 per group not served")
assert lib.wfae_set_matmul_precision(1) == 0
try:
    expect(lib.wfae_g3b_fwd_bf16(P + 2, Q, R, 32, 32, 384, 384, 8, 0, WS, big, None), SHAPE, "g3b_fwd misaligned")
    expect(lib.wfae_g3b_fwd_bf16(P, Q, R, 32, 32, 384, 384, 8, 0, WS, 64, None), WSP, "g3b_fwd short workspace")
    expect(lib.wfae_g3b_bwd_weight_bf16(P, Q, R, 32, 32, 384, 384, 8, 0, WS, 1024, None), WSP, "g3b_bwd_weight short workspace")
    for c, h in [(32, 384), (64, 192), (128, 96), (256, 48), (256, 24)]:
        for tr in (0, 1):
            expect(lib.wfae_g3b_fwd_bf16(P, Q, R, 32, c, h, h, 8, tr, WS, big, None), ANY_FAIL, f"g3b_fwd {c}@{h}")
        expect(lib.wfae_g3b_bwd_weight_bf16(P, Q, R, 32, c, h, h, 8, 0, WS, big, None), ANY_FAIL, f"g3b_bwd_weight {c}@{h}")
    assert lib.wfae_g3b_f32_supported(128, 96, 96, 8, 1) == 0   # fp32 planes only at fp32 precision
finally:
    assert lib.wfae_set_matmul_precision(0) == 0
# register-direct 1x1 convolutions (csrc/c1r.hip): shape table, strides, residual rule, stat capacity, then every served geometry up to the launch
assert lib.wfae_c1r_supported(32, 128, 384 * 384) == 1 and lib.wfae_c1r_supported(256, 64, 192 * 192) == 1
assert lib.wfae_c1r_supported(64, 256, 100) == 0 and lib.wfae_c1r_supported(256, 1024, 48 * 48) == 0 and lib.wfae_c1r_supported(128, 512, 96 * 96) == 0
assert lib.wfae_c1r_stat_rows(64, 256, 32, 192 * 192) > 0 and lib.wfae_c1r_stat_rows(64, 256, 1, 64) == 8
assert lib.wfae_c1r_stat_rows(512, 128, 32, 96 * 96) == 256 // 8 * 4 and lib.wfae_c1r_stat_rows(1024, 256, 32, 48 * 48) == 256 // 16 * 4
checked += 6
expect(lib.wfae_c1r_fwd(None, 256, 1, Q, None, None, None, R, 32, 256, 64, 192 * 192, None, 0, None, None), NULL, "c1r_fwd null")
expect(lib.wfae_c1r_fwd(P, 256, 1, Q, None, None, None, R, 32, 256, 64, 100, None, 0, None, None), UNS, "c1r_fwd HW % 64")
expect(lib.wfae_c1r_fwd(P, 1024, 1, Q, None, None, None, R, 32, 1024, 256, 48 * 48, None, 0, None, None), UNS, "c1r_fwd unserved (M, K)")
expect(lib.wfae_c1r_fwd(P, 7, 1, Q, None, None, None, R, 32, 256, 64, 192 * 192, None, 0, None, None), SHAPE, "c1r_fwd weight strides")
expect(lib.wfae_c1r_fwd(P, 256, 1, Q, None, None, S, R, 32, 256, 64, 192 * 192, None, 0, None, None), UNS, "c1r_fwd residual on a narrowing product")
expect(lib.wfae_c1r_fwd(P, 256, 1, Q + 4, None, None, None, R, 32, 256, 64, 192 * 192, None, 0, None, None), UNS, "c1r_fwd misaligned x")
expect(lib.wfae_c1r_fwd(P, 256, 1, Q, S, None, None, R, 32, 256, 64, 192 * 192, None, 0, None, None), NULL, "c1r_fwd scale without shift")
rows_c = ctypes.c_int(0)
expect(lib.wfae_c1r_fwd(P, 256, 1, Q, None, None, None, R, 32, 256, 64, 192 * 192, WS, 16, ctypes.cast(ctypes.pointer(rows_c), ctypes.c_void_p), None),
       WSP, "c1r_fwd short stat buffer")
for m, k, hw in [(32, 128, 384 * 384), (64, 256, 192 * 192), (128, 32, 384 * 384), (256, 64, 192 * 192), (512, 128, 96 * 96),
                 (1024, 256, 48 * 48), (1024, 256, 24 * 24)]:
    expect(lib.wfae_c1r_fwd(P, k, 1, Q, S, S, S if m > k else None, R, 32, k, m, hw, WS, big, ctypes.cast(ctypes.pointer(rows_c), ctypes.c_void_p), None),
           ANY_FAIL, f"c1r_fwd {k}->{m}")
    expect(lib.wfae_c1r_fwd(P, 1, m, Q, None, None, None, R, 32, k, m, hw, None, 0, None, None), ANY_FAIL, f"c1r_fwd transposed {k}->{m}")
assert lib.wfae_set_split_gemm(0) == 0
expect(lib.wfae_c1r_fwd(P, 256, 1, Q, None, None, None, R, 32, 256, 64, 192 * 192, None, 0, None, None), UNS, "c1r_fwd needs the split switch")
assert lib.wfae_set_split_gemm(1) == 0
# the same on bf16-stored tensors (csrc/c1rb.hip): precision mode, shape table, then every served geometry up to the launch
expect(lib.wfae_c1rb_fwd(P, 256, 1, Q, None, None, None, R, 32, 256, 64, 192 * 192, None, 0, None, None), UNS, "c1rb_fwd needs bf16 precision")
assert lib.wfae_c1rb_supported(64, 256, 192 * 192) == 0
assert lib.wfae_set_matmul_precision(1) == 0
try:
    assert lib.wfae_c1rb_supported(64, 256, 192 * 192) == 1 and lib.wfae_c1rb_supported(256, 1024, 24 * 24) == 0
    assert lib.wfae_c1rb_stat_rows(1024, 256, 32, 48 * 48) == 256 // 16 * 4 and lib.wfae_c1rb_stat_rows(64, 256, 1, 128) == 8
    checked += 4
    expect(lib.wfae_c1rb_fwd(P, 256, 1, Q, None, None, None, R, 32, 256, 64, 24 * 24, None, 0, None, None), UNS, "c1rb_fwd HW % 128")
    expect(lib.wfae_c1rb_fwd(P, 256, 1, Q, None, None, S, R, 32, 256, 64, 192 * 192, None, 0, None, None), UNS, "c1rb_fwd residual on a narrowing product")
    expect(lib.wfae_c1rb_fwd(P, 256, 1, Q, None, None, None, R, 32, 256, 64, 192 * 192, WS, 16, ctypes.cast(ctypes.pointer(rows_c), ctypes.c_void_p), None),
           WSP, "c1rb_fwd short stat buffer")
    for c, hh in [(128, 384), (256, 192), (512, 96), (1024, 48)]:
        for m, k in ((c // 4, c), (c, c // 4)):
            expect(lib.wfae_c1rb_fwd(P, k, 1, Q, S, S, S if m > k else None, R, 32, k, m, hh * hh, WS, big,
                                     ctypes.cast(ctypes.pointer(rows_c), ctypes.c_void_p), None), ANY_FAIL, f"c1rb_fwd {k}->{m}")
            expect(lib.wfae_c1rb_fwd(P, 1, m, Q, None, None, None, R, 32, k, m, hh * hh, None, 0, None, None), ANY_FAIL, f"c1rb_fwd transposed {k}->{m}")
finally:
    assert lib.wfae_set_matmul_precision(0) == 0
expect(lib.wfae_conv4x4s2_down(P, Q, R, 32, 256, 512, 96, 96, None), ANY_FAIL, "conv4x4s2_down")
expect(lib.wfae_conv4x4s2_wgrad(P, Q, R, 32, 256, 512, 96, 96, 0, WS, big, None), ANY_FAIL, "conv4x4s2_wgrad")
expect(lib.wfae_linear_fwd(P, Q, S, R, 32, 36864, 2048, WS, big, None), ANY_FAIL, "linear_fwd 36864->2048")
expect(lib.wfae_linear_bwd_weight_splitk(P, Q, R, 2048, 512, 512, 0, WS, big, None), ANY_FAIL, "linear_bwd_weight_splitk")
assert lib.wfae_version() == 103 and lib.wfae_workspace_bytes(1 << 24) >= (1 << 26)
print(f"asan driver: {checked} calls, no sanitizer report")
