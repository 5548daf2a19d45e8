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
        SB_HIP(hipStreamCreateWithFlags(&_stream, hipStreamNonBlocking));
        SB_HIP(hipMalloc(&_d_rec, (size_t)B_max * sizeof(tmpc_record)));
        SB_HIP(hipMalloc(&_d_all, (size_t)world * B_max * sizeof(tmpc_record)));
        SB_HIP(hipMalloc(&_d_best, (size_t)B_max * sizeof(int32_t)));
    }
    ShardedSelection::~ShardedSelection()
    {
        if (_d_rec) (void)hipFree(_d_rec);
        if (_d_all) (void)hipFree(_d_all);
        if (_d_best) (void)hipFree(_d_best);
        if (_stream) (void)hipStreamDestroy(_stream);
    }

    std::vector<int> ShardedSelection::findBestPlanner(tmpc_handle *h, int n_sets, int per_rank, const void *d_guidance_id, const void *d_weight)
    {
        const int B = n_sets * per_rank;
        if (B > _B_max || B <= 0) { std::fprintf(stderr, "ShardedSelection: batch of %d exceeds %d\n", B, _B_max); std::exit(1); }
        if (tmpc_pack_records(h, _d_rec, d_guidance_id, d_weight) || tmpc_synchronize(h)) { std::fprintf(stderr, "%s\n", tmpc_last_error(h)); std::exit(1); }
        // records of rank r land at [r][n_sets][per_rank]: the layout tmpc_select_best_records expects
        SB_NCCL(ncclAllGather(_d_rec, _d_all, (size_t)B * sizeof(tmpc_record), ncclUint8, _comm, _stream));
        SB_HIP(hipStreamSynchronize(_stream));
        if (tmpc_select_best_records(h, _d_all, _world, n_sets, per_rank, _d_best) || tmpc_synchronize(h)) { std::fprintf(stderr, "%s\n", tmpc_last_error(h)); std::exit(1); }
        std::vector<int32_t> best(n_sets);
        SB_HIP(hipMemcpy(best.data(), _d_best, (size_t)n_sets * sizeof(int32_t), hipMemcpyDeviceToHost));
        return std::vector<int>(best.begin(), best.end());
    }
}
