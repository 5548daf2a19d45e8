"""N>1 path on CPU: world_size-2 gloo processes shard a scene's trajectories, all-gather the 16-byte records and
run the deterministic selection; the result must equal the single-process FindBestPlanner.  (The per-trajectory
solves are stubbed by the CPU oracle here -- the point is the sharding / record / collective / selection logic;
on GPUs the records come from tmpc_pack_records and the collective is RCCL.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_scenes, per_scene, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from mpc_planner_amd import scenes, distributed as D
    pb = O.problem(N=20, S=5, n_lin=8, M=8, n_sqp=3)
    per_rank = per_scene // world
    recs = np.zeros((n_scenes, per_rank), D.RECORD_DTYPE)
    for s in range(n_scenes):
        sc = scenes.make_scene(40 + s, N=20, M=8, B=per_scene)
        lo, hi = D.shard_bounds(per_scene, world, rank)
        assert hi - lo == per_rank
        _, _, info = O.solve_batch(pb, sc["xinit"][lo:hi], sc["x0"][lo:hi].reshape(per_rank, -1),
                                   sc["params"][lo:hi].reshape(per_rank, -1), num_threads=2)
        recs[s] = D.pack_records_host(info["pobj"], info["exit_code"], sc["guidance_id"][lo:hi])
    local = torch.from_numpy(recs.view(np.int64).reshape(n_scenes * per_rank, 2).copy())
    gathered = D.all_gather_records(local, world)                       # [world][n_scenes*per_rank][2]
    g = gathered.numpy().reshape(world, n_scenes, per_rank, 2).copy().view(D.RECORD_DTYPE).reshape(world, n_scenes, per_rank)
    best = D.find_best_planner_records(g)
    q.put((rank, best.tolist(), g["guidance_id"][:, 0, :].tolist()))
    dist.barrier(); dist.destroy_process_group()


def test_sharded_selection_matches_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from mpc_planner_amd import scenes, distributed as D
    world, n_scenes, per_scene = 2, 3, 8
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scenes, per_scene, q)) for r in range(world)]
    for p in procs: p.start()
    outs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs: p.join(timeout=60)
    assert outs[0][1] == outs[1][1]                                     # every rank agrees without a 2nd collective
    pb = O.problem(N=20, S=5, n_lin=8, M=8, n_sqp=3)
    for s in range(n_scenes):
        sc = scenes.make_scene(40 + s, N=20, M=8, B=per_scene)
        _, _, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(per_scene, -1), sc["params"].reshape(per_scene, -1), num_threads=2)
        assert outs[0][1][s] == O.find_best(info["pobj"], info["exit_code"])
    assert outs[0][2] == [[0, 1, 2, 3], [4, 5, 6, 7]]                   # guidance ids travelled with the records


def _worker_uneven(rank, world, port, n_scenes, per_scene, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from mpc_planner_amd import scenes, distributed as D
    pb = O.problem(N=20, S=5, n_lin=8, M=8, n_sqp=2)
    lo, hi = D.shard_bounds(per_scene, world, rank)
    n_local, n_max = hi - lo, D.max_shard(per_scene, world)
    recs = np.zeros((n_scenes, n_local), D.RECORD_DTYPE)
    for s in range(n_scenes):
        sc = scenes.make_scene(60 + s, N=20, M=8, B=per_scene)
        _, _, info = O.solve_batch(pb, sc["xinit"][lo:hi], sc["x0"][lo:hi].reshape(n_local, -1),
                                   sc["params"][lo:hi].reshape(n_local, -1), num_threads=1)
        recs[s] = D.pack_records_host(info["pobj"], info["exit_code"], sc["guidance_id"][lo:hi])
    local = torch.from_numpy(recs.view(np.int64).reshape(n_scenes * n_local, 2).copy())
    padded = D.pad_records(local, n_scenes, n_local, n_max)              # uneven shards -> equal contributions
    gathered = D.all_gather_records(padded, world)
    g = gathered.numpy().reshape(world, n_scenes, n_max, 2).copy().view(D.RECORD_DTYPE).reshape(world, n_scenes, n_max)
    best = D.padded_to_global(D.find_best_planner_records(g), per_scene, world)
    q.put((rank, best.tolist(), n_local))
    dist.barrier(); dist.destroy_process_group()


def test_uneven_shards_world_size_4():
    """10 trajectories per scene over 4 ranks (shards 3, 3, 2, 2): padded to the largest shard for the one all-gather, padding
    masked by its exit code, winner mapped back to the scene's numbering; equal to the single-process selection on every rank."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from mpc_planner_amd import scenes
    world, n_scenes, per_scene = 4, 2, 10
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, n_scenes, per_scene, q)) for r in range(world)]
    for p in procs: p.start()
    outs = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs: p.join(timeout=60)
    assert [o[2] for o in outs] == [3, 3, 2, 2]
    assert all(o[1] == outs[0][1] for o in outs)
    pb = O.problem(N=20, S=5, n_lin=8, M=8, n_sqp=2)
    for s in range(n_scenes):
        sc = scenes.make_scene(60 + s, N=20, M=8, B=per_scene)
        _, _, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(per_scene, -1), sc["params"].reshape(per_scene, -1), num_threads=2)
        assert outs[0][1][s] == O.find_best(info["pobj"], info["exit_code"])


def test_padding_helpers():
    from mpc_planner_amd import distributed as D
    assert D.max_shard(10, 4) == 3 and D.max_shard(8, 4) == 2
    # padded numbering rank * 3 + t  ->  scene numbering: shards (0,3) (3,6) (6,8) (8,10)
    assert D.padded_to_global([0, 2, 3, 5, 6, 7, 9, 10, -1], 10, 4).tolist() == [0, 2, 3, 5, 6, 7, 8, 9, -1]
    t = torch.arange(2 * 2 * 2, dtype=torch.int64).reshape(4, 2)
    p = D.pad_records(t, 2, 2, 3)
    rec = p.numpy().reshape(2, 3, 2).copy().view(D.RECORD_DTYPE).reshape(2, 3)
    assert (rec["exit_code"][:, 2] == D.EMPTY_EXIT_CODE).all() and np.isinf(rec["objective"][:, 2]).all()
    assert (p.view(2, 3, 2)[:, :2] == t.view(2, 2, 2)).all()


def test_record_layout_and_host_selection():
    from mpc_planner_amd import distributed as D
    assert D.RECORD_DTYPE.itemsize == 16                                # struct tmpc_record
    rec = np.zeros((2, 2, 3), D.RECORD_DTYPE)
    rec["objective"] = [[[5, 1, 1], [9, 9, 9]], [[1, 0.5, 7], [9, 9, 9]]]
    rec["exit_code"] = [[[1, 1, 1], [4, 0, 4]], [[1, 4, 1], [4, 4, 4]]]
    assert D.find_best_planner_records(rec).tolist() == [1, -1]         # tie 1.0 -> lowest global index; failures skipped
    assert [D.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
