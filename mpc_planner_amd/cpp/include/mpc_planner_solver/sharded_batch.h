/*
 * mpc_planner_solver/sharded_batch.h -- native (C++) multi-GPU path of the batched planners, for a host process
 * per GPU that does not go through Python / torch.distributed: a guidance (or scenario) set is split in contiguous blocks over
 * the ranks, every rank solves its block with one launch, packs one 16-byte record per local trajectory, ONE RCCL all-gather
 * over xGMI moves the records (world x B x 16 bytes: latency-bound, one-shot), and every rank runs the same deterministic
 * FindBestPlanner (guidance_constraints.cpp:416-434: lowest global index among exit_code == 1 with the smallest objective) on the
 * gathered array -- no second collective.  The library itself (libtmpc_hip.so) makes no RCCL call; this helper links librccl.
 */
#ifndef MPC_PLANNER_HIP_SHARDED_BATCH_H
#define MPC_PLANNER_HIP_SHARDED_BATCH_H

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <vector>

#include "tmpc_hip.h"

namespace MPCPlanner
{
    class ShardedSelection
    {
    public:
        /* comm: the ranks that share the planner sets; B_max: local trajectories per launch */
        ShardedSelection(ncclComm_t comm, int rank, int world, int B_max);
        ~ShardedSelection();
        ShardedSelection(const ShardedSelection &) = delete;
        /* Stream-ordered on the handle's own stream (pack -> ncclAllGather -> selection -> copy of the winners), one host wait at the end.
         * After tmpc_solve / tmpc_solve_iterations on `h` (B = n_sets * per_rank local trajectories, set s = trajectories
         * [s per_rank, (s+1) per_rank)): returns per set the winner's GLOBAL index rank * per_rank + t (or -1).
         * d_guidance_id / d_weight: device arrays [B] or nullptr (tmpc_pack_records). */
        std::vector<int> findBestPlanner(tmpc_handle *h, int n_sets, int per_rank, const void *d_guidance_id = nullptr, const void *d_weight = nullptr);
        const tmpc_record *gatheredRecordsDevice() const { return (const tmpc_record *)_d_all; }

    private:
        ncclComm_t _comm;
        int _rank, _world, _B_max;
        void *_d_rec{nullptr}, *_d_all{nullptr}, *_d_best{nullptr};
        std::vector<int32_t> _host_best;
        int32_t *_h_best_scratch(int n);
    };
}
#endif
