// Native multi-GPU selection (mpc_planner_solver/sharded_batch.h) on a one-rank RCCL communicator: the all-gather + gathered
// FindBestPlanner path a C++ host uses, checked against tmpc_select_best on the same launch.  Multi-rank runs need the node the
// driver has; the record layout [rank][set][per_rank] and the global-index rule are covered on CPU by tests/test_distributed_gloo.py.
#include <mpc_planner_solver/sharded_batch.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace MPCPlanner;
#define ASSERT_TRUE(c) do { if (!(c)) { std::printf("ASSERT FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");                 // [B, N, npar, nx, nvar] then xinit, x0, params (doubles)
    if (!f) return 2;
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<double> in(n / 8);
    if (std::fread(in.data(), 8, in.size(), f) != in.size()) return 2;
    std::fclose(f);
    const int B = (int)in[0], N = (int)in[1], n_sets = (int)in[2], per = B / n_sets;
    tmpc_dims d; tmpc_default_dims(&d, N, 5, 8, 8);
    const double *xinit = &in[3], *x0 = xinit + (size_t)B * 5, *params = x0 + (size_t)B * (N + 1) * 7;
    tmpc_handle *h = nullptr;
    ASSERT_TRUE(tmpc_create(&h, &d, B, 0) == 0);
    ASSERT_TRUE(tmpc_set_batch(h, B, xinit, x0, params) == 0 && tmpc_solve(h) == 0);
    ncclUniqueId id; ncclComm_t comm;
    ASSERT_TRUE(ncclGetUniqueId(&id) == ncclSuccess && ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess);
    {
        ShardedSelection sel(comm, 0, 1, B);
        std::vector<int> best = sel.findBestPlanner(h, n_sets, per);
        for (int s = 0; s < n_sets; s++) {
            int32_t ref = -2;
            ASSERT_TRUE(tmpc_select_best(h, s * per, per, nullptr, nullptr, &ref) == 0);
            ASSERT_TRUE(best[s] == ref);
            std::printf("set %d best %d\n", s, best[s]);
        }
    }
    ncclCommDestroy(comm);
    tmpc_destroy(h);
    std::printf("sharded ok\n");
    return 0;
}
