"""Host-side logic of the grouped weight gradients (no GPU): the library's K-range choice / time model and the partition of
a stack of residual blocks into groups (ops.wgrad_group_plan).  Shapes: reference modules/module_seg_vit.py:162-196 (ViT-B/16:
D = 768, 4D MLP, 256 x 196 token rows) and modules/module_clip_ttransformer.py:20-37 (text: D = 512, 256 x 77 rows)."""
import ctypes as C

import pytest

from segclip_amd import _lib as L
from segclip_amd import ops


def tiles_per_block(D, F4):
    return (4 * D * D + 2 * F4 * D) // 65536


def test_k_range_choice_and_model():
    lib = L.load()
    # one vision gradient alone (27 tiles of 256 x 256 over 784 K steps) needs many ranges to occupy 256 CUs ...
    assert lib.segclip_wgrad_group_splits(27, 784) >= 7
    # ... seven blocks together (756 tiles = 2.95 rounds) need none, and no partial tile is written
    assert lib.segclip_wgrad_group_splits(7 * tiles_per_block(768, 3072), 784) == 1
    # an empty K range is never chosen, and the model grows with the work
    for tiles, ksteps in ((1, 9), (4, 16), (48, 308), (1296, 784)):
        s = lib.segclip_wgrad_group_splits(tiles, ksteps)
        assert 1 <= s <= 32 and (s - 1) * -(-ksteps // s) < ksteps
        assert lib.segclip_wgrad_group_model_us(2 * tiles, ksteps, s) >= lib.segclip_wgrad_group_model_us(tiles, ksteps, s) > 0.0
    assert lib.segclip_wgrad_group_splits(0, 0) == 1


@pytest.mark.parametrize("nblk,D,F4,rows", [(10, 768, 3072, 50176), (2, 768, 3072, 50176), (12, 512, 2048, 19712), (23, 1024, 4096, 73728)])
def test_partition_of_a_stack_into_groups(nblk, D, F4, rows):
    lib = L.load()
    tb = tiles_per_block(D, F4)
    for gmax in (1, 3, 12):
        plan = ops.wgrad_group_plan(nblk, tb, rows // 64, gmax)
        assert sum(plan) == nblk and all(1 <= n <= min(gmax, 12) for n in plan)
        assert plan == ops.wgrad_group_plan(nblk, tb, rows // 64, gmax)          # deterministic

        def cost(p):
            return sum(lib.segclip_wgrad_group_model_us(tb * n, rows // 64, lib.segclip_wgrad_group_splits(tb * n, rows // 64)) for n in p)
        assert cost(plan) <= cost([1] * nblk) + 1e-6                               # never worse than one launch per block
    if (nblk, D) == (10, 768):
        assert ops.wgrad_group_plan(nblk, tb, rows // 64, 12) == [7, 3]           # the plan DESIGN 4.5 quotes


def test_group_entry_rejects_what_it_does_not_cover_without_a_gpu():
    """Argument checks happen before any launch: n = 0 / K not a multiple of 64 / 128-column problems are UNSUPPORTED."""
    lib = L.load()
    it = (L.WgradItem * 1)()
    it[0].dy, it[0].x, it[0].dw = 256, 512, 1024          # fake, 16-byte aligned addresses: never dereferenced on this path
    it[0].M, it[0].N, it[0].ld_dy, it[0].ld_x, it[0].ld_dw = 128, 256, 128, 256, 256
    assert lib.segclip_wgrad_group(it, 0, 6272, 1, None, 0, None) == -2
    assert lib.segclip_wgrad_group(it, 1, 100, 1, None, 0, None) == -2
    assert lib.segclip_wgrad_group(it, 1, 6272, 1, None, 0, None) == -2      # M = 128 is not a multiple of 256
    assert b"wgrad_group" in lib.segclip_last_error_string()


def test_half_tile_tail_plan():
    """gemm_bf16_pq.hip: the tiles of a last round that is at most half full run as 128 x 256 workgroups"""
    lib = L.load()
    assert lib.segclip_gemm_pq_half_tail(197 * 3) == 79        # N = 768 at 256 x 197 rows: 2 full rounds + 79 tiles
    assert lib.segclip_gemm_pq_half_tail(197 * 12) == 60       # N = 3072: 9 rounds + 60
    assert lib.segclip_gemm_pq_half_tail(197 * 9) == 0         # N = 2304: the last round has 237 tiles
    assert lib.segclip_gemm_pq_half_tail(256 + 128) == 128 and lib.segclip_gemm_pq_half_tail(256 + 129) == 0
    for n in (0, 1, 154, 256, 512):                            # one partial round only / whole rounds: nothing to split
        assert lib.segclip_gemm_pq_half_tail(n) == 0
