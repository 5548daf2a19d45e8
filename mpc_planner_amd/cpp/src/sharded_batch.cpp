// Native multi-GPU selection over RCCL -- see include/mpc_planner_solver/sharded_batch.h.
#include <mpc_planner_solver/sharded_batch.h>

#include <cstdio>
#include <cstdlib>

namespace MPCPlanner
{
#define SB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define SB_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); std::exit(1); } } while (0)

    ShardedSelection::ShardedSelection(ncclComm_t comm, int rank, int world, int B_max) : _comm(comm), _rank(rank), _world(world), _B_max(B_max)
    {
        SB_HIP(hipMalloc(&_d_rec, (size_t)B_max * sizeof(tmpc_record)));
        SB_HIP(hipMalloc(&_d_all, (size_t)world * B_max * sizeof(tmpc_record)));
        SB_HIP(hipMalloc(&_d_best, (size_t)B_max * sizeof(int32_t)));
    }
    ShardedSelection::~ShardedSelection()
    {
        if (_d_rec) (void)hipFree(_d_rec);
        if (_d_all) (void)hipFree(_d_all);
        if (_d_best) (void)hipFree(_d_best);
    }

    std::vector<int> ShardedSelection::findBestPlanner(tmpc_handle *h, int n_sets, int per_rank, const void *d_guidance_id, const void *d_weight)
    {
        const int B = n_sets * per_rank;
        if (B > _B_max || B <= 0) { std::fprintf(stderr, "ShardedSelection: batch of %d exceeds %d\n", B, _B_max); std::exit(1); }
        // pack -> all-gather -> select, all enqueued on the HANDLE's stream (tmpc_get_stream): RCCL waits for the packed records and
        // the selection kernel for RCCL by stream order -- no host synchronisation between them (round 2 had two)
        void *st = nullptr;
        if (tmpc_pack_records(h, _d_rec, d_guidance_id, d_weight) || tmpc_get_stream(h, &st)) { std::fprintf(stderr, "%s\n", tmpc_last_error(h)); std::exit(1); }
        // records of rank r land at [r][n_sets][per_rank]: the layout tmpc_select_best_records expects
        SB_NCCL(ncclAllGather(_d_rec, _d_all, (size_t)B * sizeof(tmpc_record), ncclUint8, _comm, (hipStream_t)st));
        if (tmpc_select_best_records(h, _d_all, _world, n_sets, per_rank, _d_best)) { std::fprintf(stderr, "%s\n", tmpc_last_error(h)); std::exit(1); }
        SB_HIP(hipMemcpyAsync(_h_best_scratch(n_sets), _d_best, (size_t)n_sets * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)st));
        if (tmpc_synchronize(h)) { std::fprintf(stderr, "%s\n", tmpc_last_error(h)); std::exit(1); }      // the one wait: the winners are needed on the host
        return std::vector<int>(_host_best.begin(), _host_best.begin() + n_sets);
    }
    int32_t *ShardedSelection::_h_best_scratch(int n) { if ((int)_host_best.size() < n) _host_best.resize(n); return _host_best.data(); }
}
