"""N>1 path on CPU with the gloo backend, world_size 2 (no GPU needed).

Each rank plays one GPU: it produces the dense gradients of ITS view (computed here by the CPU oracle, since the
HIP op has no CPU path).  Both exchange forms of brush_amd.dist run over gloo: the dense all-reduce of the block
prefix, and the ViewExchange of 64-byte records (sizing, exact-size all-gather, per-splat sum in view order through
the torch restatements of the HIP kernels) which must give the same sum AND the same bits on both ranks.  Also
checks the block layout, view sharding and the densification statistics.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _view_uniforms(rank, w, h, deg):
    c = H.reference_test_camera(w, h)
    pos = [0.4 * rank, -0.2 * rank, -8.0]
    return O.make_uniforms(pos, c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"], [w, h], deg)


def _oracle_block(rank, cloud, w, h, deg):
    """Gradient block of view `rank` laid out like brush_amd.render.grad_block_layout."""
    from brush_amd.render import grad_block_layout

    n = cloud["means"].shape[0]
    ncoef = (deg + 1) ** 2
    u = _view_uniforms(rank, w, h, deg)
    out, aux = O.render_forward(u, cloud["means"], cloud["log_scales"], cloud["quats"], cloud["sh"], cloud["raw_opac"])
    v_out = np.full((h, w, 4), 1.0 / (4 * w * h), np.float32)
    g = O.render_backward(u, aux, cloud["means"], cloud["log_scales"], cloud["quats"], cloud["raw_opac"], out, v_out)
    layout, total = grad_block_layout(n, ncoef)
    block = np.zeros(total, np.float32)
    for name, key in (("v_means", "v_means"), ("v_scales", "v_scales"), ("v_quats", "v_quats"), ("v_opac", "v_opac"),
                      ("v_sh", "v_sh"), ("v_xy", "v_xy")):
        off, sz = layout[name]
        block[off:off + sz] = g[key].reshape(-1)
    return block, aux, g


def _inverse_map(aux_np, n):
    inv = np.full(n, -1, np.int32)
    V = int(aux_np["num_visible"][0])
    inv[aux_np["global_from_compact_gid"][:V]] = np.arange(V, dtype=np.int32)
    return torch.from_numpy(inv)


def _worker(rank, world, port, n, w, h, deg, packed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from brush_amd import dist as BD
        from brush_amd.render import RenderAux, grad_block_layout

        cloud = H.synthetic_cloud(n, deg, seed=4, mean_mult=0.002)
        block_np, aux_np, g = _oracle_block(rank, cloud, w, h, deg)
        block = torch.from_numpy(block_np.copy())
        ncoef = (deg + 1) ** 2
        BD.allreduce_param_grads(block, n, ncoef)
        # densification statistics of this view
        as_i32 = lambda a: torch.from_numpy(a.view(np.int32).copy())
        aux = RenderAux(projected_splats=torch.from_numpy(aux_np["projected_splats"]), uniforms_buffer=torch.zeros(28, dtype=torch.int32),
                        num_intersections=as_i32(aux_np["num_intersections"]), num_visible=as_i32(aux_np["num_visible"]),
                        final_index=as_i32(aux_np["final_index"]), cum_tiles_hit=as_i32(aux_np["cum_tiles_hit"]),
                        tile_bins=as_i32(aux_np["tile_bins"]), compact_gid_from_isect=as_i32(aux_np["compact_gid_from_isect"]),
                        global_from_compact_gid=as_i32(aux_np["global_from_compact_gid"]),
                        compact_from_global_gid=_inverse_map(aux_np, n), overflow=torch.zeros(1, dtype=torch.int32))
        # record exchange: all-gather of 64-byte records of the visible splats + per-splat sum in view order
        ub = torch.zeros(28, dtype=torch.int32)
        ub[:16] = torch.from_numpy(_view_uniforms(rank, w, h, deg)["viewmat"].view(np.int32).copy())
        aux.uniforms_buffer = ub
        xchg = BD.ViewExchange(n, ncoef, torch.device("cpu"), packed=packed)
        xchg.begin(aux)
        rows = -(-max(xchg.counts()) // 256) * 256
        dense = {k: torch.from_numpy(g[k]) for k in ("v_means", "v_scales", "v_quats", "v_opac", "v_sh", "v_xy")}
        xchg.set_local_records(BD.records_from_dense_torch(dense, aux, n, (w, h), rows))
        recs = xchg.gather()
        red = BD.reduce_view_records_torch(recs, xchg.metas[:, 0], xchg.metas[:, 1:4].contiguous().view(torch.float32),
                                           torch.from_numpy(cloud["means"]), n, ncoef)
        stats = BD.densification_stats(torch.from_numpy(g["v_xy"]), aux, (w, h))
        local_stats = stats.clone()
        BD.allreduce_densification_stats(stats)
        q.put((rank, block.numpy(), block_np, local_stats.numpy(), stats.numpy(), int(aux_np["num_visible"][0]),
               {k: v.numpy() for k, v in red.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("packed", [False, True])
def test_view_sharded_allreduce_gloo_world2(packed):
    from brush_amd import dist as BD
    from brush_amd.render import grad_block_layout

    n, w, h, deg, world = 3000, 96, 64, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, w, h, deg, packed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=240)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ncoef = (deg + 1) ** 2
    prefix = BD.param_grad_floats(n, ncoef)
    layout, total = grad_block_layout(n, ncoef)
    assert prefix == layout["v_sh"][0] + layout["v_sh"][1] and prefix < total
    want = res[0][1][:prefix] + res[1][1][:prefix]
    for r in range(world):
        reduced, local = res[r][0], res[r][1]
        assert np.array_equal(reduced[:prefix], want), "parameter gradients must be the sum over views"
        assert np.array_equal(reduced[prefix:], local[prefix:]), "v_xy (per-view statistic) must not be reduced"
        # the two views differ, so the reduce really changed something
        assert not np.array_equal(reduced[:prefix], local[:prefix])
    # the record exchange gives the same dense sum (one extra rounding in v_rgb = v_sh0 / Y0 aside) ...
    layout, _ = grad_block_layout(n, ncoef)
    for r in range(world):
        red = res[r][5]
        for name in ("v_means", "v_scales", "v_quats", "v_opac", "v_sh"):
            off, sz = layout[name]
            a, b = red[name].reshape(-1).astype(np.float64), want[off:off + sz].astype(np.float64)
            scale = np.abs(b).max() + 1e-30
            assert np.abs(a - b).max() <= 2e-6 * scale, (name, np.abs(a - b).max(), scale)
    # ... and the SAME BITS on every rank (replicated parameters must stay replicated), statistics included
    for name in res[0][5]:
        assert np.array_equal(res[0][5][name].view(np.uint32), res[1][5][name].view(np.uint32)), name
    # densification stats: sums over views; visibility row counts views in which a splat is visible
    s_sum = res[0][2] + res[1][2]
    assert np.allclose(res[0][3], s_sum) and np.allclose(res[1][3], s_sum)
    # the records carry the same statistics (|v_xy * (w/2, h/2)| per view, views that saw the splat)
    assert np.allclose(res[0][5]["xy_norm"], s_sum[0], rtol=1e-6, atol=1e-12)
    assert np.array_equal(res[0][5]["views_seen"], s_sum[1])
    for r in range(world):
        assert int(res[r][2][1].sum()) == res[r][4]  # row 1 of the local stats marks exactly V splats
    assert BD.shard_views(8, 1, 2) == [1, 3, 5, 7] and BD.shard_views(3, 2, 4) == [2]


def test_block_layout_is_contiguous_and_aligned():
    from brush_amd.render import grad_block_layout

    for n, c in ((1, 1), (7, 16), (1000, 25), (1 << 20, 16)):
        layout, total = grad_block_layout(n, c)
        order = ["v_means", "v_scales", "v_quats", "v_opac", "v_sh", "v_xy"]
        end = 0
        for name in order:
            off, sz = layout[name]
            assert off % 4 == 0 and off >= end
            end = off + sz
        assert total >= end
