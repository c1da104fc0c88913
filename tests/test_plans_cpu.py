"""Host-side launch plans vs the workspace-size queries (no GPU): for every shape swept, the extent an entry point
writes must be no larger than the size the caller was told to allocate, and every split / chunk must start inside
the matrix.  Covers the planning code a layout-dependent out-of-bounds access could hide in
(include/tp3d_hip.h, "Launch plans")."""
import ctypes
import itertools

import pytest

from torch_points3d_amd import _lib


def _plan(name, n, *args):
    buf = (ctypes.c_int64 * n)()
    rc = getattr(_lib.load(), name)(*args, ctypes.addressof(buf))
    assert rc == 0, (name, args, rc)
    return list(buf)


# rows of the grouped-MLP layers: every golden fixture's shapes, BASELINE configs 1-3, and the edges of the plans
ROWS = sorted(set(
    [1, 2, 3, 31, 32, 33, 63, 64, 65, 120, 127, 128, 129, 255, 256, 257, 480, 512, 1000, 1023, 1024, 1025, 2048, 2100,
     3840, 4095, 4096, 4097, 8191, 8192, 11520, 16384, 65535, 65536, 65537, 130047, 130048, 130049, 130176, 130177,
     130944, 131071, 131072, 131073, 262144, 524288, 1048575, 1048576, 1048577, 2097152, 4194304]
    + [128 * k + d for k in (1016, 1017, 1020, 1023, 1024, 1025) for d in (-1, 0, 1)]))
CHANNELS = [1, 3, 4, 6, 8, 10, 13, 16, 20, 24, 32, 48, 50, 64, 96, 128, 131, 132, 192, 196, 256, 259, 260, 320, 323, 384,
            512, 515, 516, 1024, 1280, 1536]


def test_gemm_tn_plan_fits_its_workspace():
    h = _lib.load()
    for M in ROWS:
        for N, K in itertools.product(CHANNELS, CHANNELS):
            if N * K > 1024 * 1280:
                continue
            splits, rps, tn, tk, tiles, staged, floats, last_start = _plan("tp3d_gemm_tn_plan", 8, M, N, K)
            ws = h.tp3d_gemm_tn_workspace_floats(M, N, K)
            assert floats == splits * N * K <= ws, (M, N, K)
            assert 1 <= splits <= 512 and rps % 64 == 0 and rps % staged == 0, (M, N, K, splits, rps, staged)
            assert last_start < M and splits * rps >= M, (M, N, K, splits, rps)  # no empty split, rows all covered
            assert tiles * tn * tk >= N * K


def test_gemm_tn_x3_plan_fits_its_workspace_and_lds():
    """the bf16-pipe weight-gradient kernel (csrc/gemm_tn_x3.hip): shapes it declares served, its partial-tile extent, its
    dynamic LDS (must fit the CU's 160 KiB), the 32-bit byte offsets its loader lanes use"""
    h = _lib.load()
    served = 0
    for M in ROWS:
        for N, K in itertools.product(CHANNELS, CHANNELS):
            ok = h.tp3d_gemm_tn_x3_serves(M, N, K)
            base = M >= 131072 and N >= 64 and K >= 64 and N % 4 == 0 and K % 4 == 0 and M * max(N, K) < 2 ** 30
            assert not ok or base, (M, N, K)
            if not ok:
                assert h.tp3d_gemm_tn_x3_workspace_floats(M, N, K) == 0
                continue
            served += 1
            splits, rps, tn, tk, tiles, staged, floats, lds = _plan("tp3d_gemm_tn_x3_plan", 8, M, N, K)
            assert floats == splits * N * K == h.tp3d_gemm_tn_x3_workspace_floats(M, N, K), (M, N, K)
            assert 1 <= splits <= 256 and tiles <= 2 and splits * tiles <= 256, (M, N, K, splits, tiles)
            assert N * K * 5 >= tiles * tn * tk * 4, (M, N, K, tn, tk)  # at least 80 % of the tiles is output
            assert rps % staged == 0 and splits * rps >= M and (splits - 1) * rps < M, (M, N, K, splits, rps)
            assert lds <= 160 * 1024 and staged in (32, 64), (M, N, K, lds)
            assert tn in (64, 128) and tk in (64, 128, 160) and tiles * tn * tk >= N * K, (M, N, K, tn, tk, tiles)
            if 128 < K <= 160 and N > 64:
                assert tk == 160 and tiles == -(-N // 128)  # the strip: dY is read once, not once per tile column
            assert M * max(N, K) * 4 < 2 ** 32  # loader lanes address a row block with 32-bit byte offsets
    for shape in ((524288, 128, 128), (524288, 128, 132), (1048576, 128, 64), (262144, 256, 128), (1048576, 64, 64)):  # the BASELINE step's layers
        assert h.tp3d_gemm_tn_x3_serves(*shape), shape
    for shape in ((524288, 256, 196), (2097152, 128, 96), (524288, 128, 324), (40000, 128, 128)):
        assert not h.tp3d_gemm_tn_x3_serves(*shape), shape
    assert served > 100


def test_gemm_rows_plan_fits_its_statistics_buffer():
    h = _lib.load()
    for M in ROWS:
        for N in CHANNELS:
            tiles_n, row_blocks, items, blocks, chunks, per_wg, wave_rows, _, bn = _plan("tp3d_gemm_rows_plan", 9, M, N, 0)
            floats = h.tp3d_gemm_rows_stat_floats(M, N)
            assert chunks == h.tp3d_gemm_rows_stat_chunks(M, N)
            assert chunks * 4 * N <= floats, (M, N)  # per chunk: sum d, sum d^2, shift, rows
            assert bn in (64, 128) and wave_rows == (2 if bn == 128 else 4) and tiles_n * bn >= N > (tiles_n - 1) * bn
            assert row_blocks * 128 >= M and items >= row_blocks * tiles_n and blocks <= 1024
            if per_wg:
                # statistics chunks of workgroup w: wave_rows * ((w // (8*tiles_n)) * 8 + (w & 7)) + wave row
                assert blocks == 1024 and 1024 % (8 * tiles_n) == 0
                top = max(wave_rows * ((w // (8 * tiles_n)) * 8 + (w & 7)) + wave_rows - 1 for w in range(blocks))
                assert top == chunks - 1
            else:
                assert chunks == wave_rows * row_blocks  # one per (128-row block, wave row)


def test_gemm_rows_k_split_fits_its_slabs():
    h = _lib.load()
    for M in ROWS:
        for N, K in itertools.product(CHANNELS, [4, 32, 128, 256, 260, 384, 512, 516, 1024, 1280, 1536]):
            if M * N > (1 << 28):
                continue
            ksplit = _plan("tp3d_gemm_rows_plan", 9, M, N, K)[7]
            floats = h.tp3d_gemm_rows_workspace_floats(M, N, K)
            assert (floats == 0) == (ksplit == 1), (M, N, K)
            if ksplit > 1:
                assert floats == ksplit * M * N and 2 <= ksplit <= 16
                kchunk = -(-(-(-K // ksplit)) // 32) * 32
                assert (ksplit - 1) * kchunk < K <= ksplit * kchunk  # every K-range starts inside the contraction


@pytest.mark.parametrize("pooled_ns", [0, 1, 2, 16, 24, 32, 64, 128, 512])
def test_bn_plans_fit_the_bn_workspace(pooled_ns):
    h = _lib.load()
    for M in ROWS:
        if pooled_ns > 1 and M % pooled_ns:
            continue
        for C in (1, 4, 10, 64, 131, 1024):
            crow, chunks, floats = _plan("tp3d_bn_plan", 3, M, C, pooled_ns)
            assert floats <= h.tp3d_bn_workspace_floats(M, C), (M, C, pooled_ns)
            assert chunks <= 65535 or M > 65535 * 256
            R = M // pooled_ns if pooled_ns > 1 else M
            assert (chunks - 1) * crow < R <= chunks * crow


def test_scatter_plan_carves_disjoint_ranges_large_enough():
    h = _lib.load()
    shapes = [(1, 1, 1), (2, 8192, 128), (2, 32768, 512), (32, 32768, 16384), (32, 8192, 512), (32, 49152, 512),
              (2, 3840, 700), (3, 960, 160), (1, 16384, 4096), (1, 16383, 4096), (1, 200000, 65536), (4, 65536, 100),
              (4, 65537, 100), (2, 90112, 512), (32, 90112, 16384), (1, 3, 70000), (2, 76800, 300), (2, 5120, 2000),
              (1, 1048576, 262144), (8, 1, 1), (1, 65536, 36000), (1, 65536, 20000)]
    for (B, L, nbins), ww in itertools.product(shapes, (0, 1)):
        o_start, o_order, o_scratch, o_w, o_merge, total, flat, flat_ints, o_hubs = _plan("tp3d_scatter_plan", 9, B, L, nbins, ww)
        assert total == h.tp3d_scatter_workspace_bytes(B, L, nbins, ww)
        ends = [(o_start, B * (nbins + 1) * 4), (o_order, B * L * 4), (o_scratch, B * L * 4)]
        if ww:
            ends.append((o_w, B * L * 4))
        else:
            assert o_w == -1
        ends.append((o_merge, B * L * 4))
        ends.append((o_hubs, (B * nbins + 1) * 4))  # count + ids of the destinations with long runs
        pos = 0
        for off, size in ends:
            assert off % 16 == 0 and off >= pos, (B, L, nbins, ww)
            pos = off + size
        assert pos <= total
        if flat:
            assert flat_ints <= B * L, (B, L, nbins)        # histogram + cursors live in `scratch`
            assert B * nbins + 1 <= B * (nbins + 1)          # flat start table inside `start`


def test_split_role_gemm_shape_rules():
    """tp3d_gemm_rows_sp_chunks / tp3d_gemm_rows_bnbwd_sp_serves (host arithmetic): which (M, N, K) the split-role GEMMs
    take, and that the statistics chunk count is what the launch writes -- 2 wave rows x workgroups per column tile, with
    every workgroup staying on one column tile (workgroups % (8 * column tiles) == 0)."""
    h = _lib.load()
    for M in ROWS:
        for N, K in itertools.product(CHANNELS, CHANNELS):
            row_blocks = (M + 127) // 128
            tiles_n = 1 if N <= 64 else (N + 127) // 128
            rem = N % 128
            items = (row_blocks + 7) // 8 * 8 * tiles_n
            fits = (K >= 4 and K % 4 == 0 and not (N > 64 and 0 < rem <= 64) and 512 % (8 * tiles_n) == 0 and items >= 512)
            for side in (0, 1):
                chunks = h.tp3d_gemm_rows_sp_chunks(M, N, K, side)
                if not (fits and K <= 512):
                    assert chunks == 0, (M, N, K, side, chunks)
                    continue
                grid = 1024 if (side and items >= 2048) else 512
                assert chunks == 2 * grid // tiles_n and grid % (8 * tiles_n) == 0 and grid <= items, (M, N, K, side, chunks)
            assert h.tp3d_gemm_rows_bnbwd_sp_serves(M, N, K) == int(fits and K <= 256), (M, N, K)
