"""Multi-GPU sharding of the T-MPC trajectory batch.

Every trajectory is an independent NLP; only `FindBestPlanner` (guidance_constraints.cpp:416-434) couples
them.  A scene's guidance trajectories are split in contiguous blocks over the ranks (one process per GPU);
after the local batched solve each rank packs one 16-byte record {f64 objective, i32 exit_code, i32 guidance_ID}
per trajectory, ONE all-gather moves the records (RCCL over xGMI with backend "nccl"; gloo in the CPU tests),
and every rank runs the same deterministic selection, so all ranks agree on the winner without a second
collective.  No other data-path collective exists.
"""
import numpy as np

RECORD_DTYPE = np.dtype([("objective", "<f8"), ("exit_code", "<i4"), ("guidance_id", "<i4")])   # = tmpc_record


def shard_bounds(total, world_size, rank):
    """Contiguous block partition of `total` trajectories; the first (total % world_size) ranks get one more."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(total, world_size):
    """Size of the largest shard of shard_bounds(total, world_size, .): what uneven shards are padded to for the collective."""
    return -(-total // world_size)


EMPTY_EXIT_CODE = -999          # EXIT_CODE_NOT_OPTIMIZED_YET (controller_module.h:13): a padding record is never eligible


def pad_records(local_records_tensor, n_sets, n_local, n_max):
    """Uneven shards: all-gather needs equal contributions, so a rank whose shard has n_local < n_max trajectories per set appends
    (n_max - n_local) records per set that can never win (exit code -999, objective +inf).  local_records_tensor: int64
    [n_sets * n_local][2]; returns [n_sets * n_max][2]."""
    import torch
    if n_local == n_max:
        return local_records_tensor
    pad = np.zeros((n_sets, n_max - n_local), RECORD_DTYPE)
    pad["objective"] = np.inf; pad["exit_code"] = EMPTY_EXIT_CODE; pad["guidance_id"] = -1
    pad_t = torch.from_numpy(pad.view(np.int64).reshape(n_sets, n_max - n_local, 2).copy()).to(local_records_tensor.device)
    return torch.cat([local_records_tensor.view(n_sets, n_local, 2), pad_t], dim=1).reshape(n_sets * n_max, 2).contiguous()


def padded_to_global(best, total, world_size):
    """Index in the padded numbering (rank * max_shard + t, what the selection over padded records returns) -> index in the
    scene's own numbering (shard_bounds(rank).lo + t); -1 stays -1.  The map is monotone, so 'lowest index wins ties' is kept."""
    best = np.asarray(best)
    n_max = max_shard(total, world_size)
    rk, t = np.divmod(np.maximum(best, 0), n_max)
    lo = np.array([shard_bounds(total, world_size, int(r))[0] for r in rk.ravel()]).reshape(rk.shape)
    return np.where(best < 0, -1, lo + t).astype(np.int32)


def pack_records_host(pobj, exit_code, guidance_id=None, weight=None):
    """Host mirror of tmpc_pack_records (SolverResult bookkeeping, guidance_constraints.cpp:344-360)."""
    rec = np.zeros(len(pobj), RECORD_DTYPE)
    rec["objective"] = pobj if weight is None else np.asarray(pobj) * np.asarray(weight)
    rec["exit_code"] = exit_code
    rec["guidance_id"] = np.arange(len(pobj)) if guidance_id is None else guidance_id
    return rec


def find_best_planner_records(records):
    """Host FindBestPlanner over gathered records [n_ranks][n_scenes][per_rank]: per scene the lowest global
    index (rank*per_rank + t) with exit_code == 1 and the smallest objective (init 1e10, strict '<')."""
    n_ranks, n_scenes, per_rank = records.shape
    best = np.full(n_scenes, -1, np.int32)
    for s in range(n_scenes):
        best_solution = 1e10
        for rk in range(n_ranks):
            for t in range(per_rank):
                r = records[rk, s, t]
                if r["exit_code"] == 1 and r["objective"] < best_solution:
                    best_solution = r["objective"]; best[s] = rk * per_rank + t
    return best


def all_gather_records(local_records_tensor, world_size):
    """local_records_tensor: torch int64 tensor [n_local][2] viewing the 16-byte records (CPU for gloo, CUDA for
    RCCL).  Returns [world_size][n_local][2]."""
    import torch
    import torch.distributed as dist
    if world_size == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_records_tensor.unsqueeze(0)
    n = local_records_tensor.shape[0]
    out = torch.empty((world_size * n,) + tuple(local_records_tensor.shape[1:]), dtype=local_records_tensor.dtype,
                      device=local_records_tensor.device)
    dist.all_gather_into_tensor(out, local_records_tensor.contiguous())      # one collective: concat along dim 0
    return out.view((world_size, n) + tuple(local_records_tensor.shape[1:]))
