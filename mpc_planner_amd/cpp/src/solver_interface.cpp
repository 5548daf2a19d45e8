// HIP flavour of MPCPlanner::Solver -- see include/mpc_planner_solver/solver_interface.h.
// Each method cites the reference method it mirrors (mpc_planner_solver/src/acados_solver_interface.cpp).
#include <mpc_planner_solver/solver_interface.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>

#ifndef SOLVER_COST_MODEL      // (dims headers written before round 4)
#define SOLVER_COST_MODEL 0
#endif
#ifndef SOLVER_ROW_MODEL
#define SOLVER_ROW_MODEL 0
#endif

namespace MPCPlanner
{
    /* Kernel variant of a control tick (tmpc_set_latency_mode).  A tick of a few planners is one dependent chain deep, and for that the
     * latency variants are built: 3 = four waves per trajectory (round 6: the stage evaluation, the row passes and the wide phases of the
     * parallel-in-time Newton solve on 256 lanes; N <= 20), 2 = the interior-point Newton systems solved parallel in time on two waves (exit
     * codes and iteration counts equal to the oracle's on every set tried, steps equal to ~1e-6, see include/tmpc_hip.h), 1 = two waves per
     * trajectory with the same stage-by-stage Riccati recursion as acados / HPIPM (rounding-equal to the throughput kernels), 0 = the
     * throughput kernels themselves.  MPC_PLANNER_HIP_TICK_VARIANT picks one (default 3).  A variant is asked for ONLY while the batch fits
     * its resident set (tmpc_latency_mode_capacity: one workgroup per CU for variants 2 and 3) and the shape has it; otherwise the next lower
     * one is tried: a larger solveBatch() runs on the throughput kernels (round-4 advisor: variant 2 for every batch size was a throughput
     * regression for large batches).  The rule looks at the batch size the CALLER handed over, so it is the caller's choice, not the
     * library's: a given batch always gets the same kernels. */
    static int tickKernelVariant()
    {
        const char *v = std::getenv("MPC_PLANNER_HIP_TICK_VARIANT");
        if (v && v[0] >= '0' && v[0] <= '3' && v[1] == '\0') return v[0] - '0';
        return 3;
    }
    static void applyTickVariant(tmpc_handle *h, int batch)
    {
        int want = tickKernelVariant();
        while (want > 0) {
            const int cap = tmpc_latency_mode_capacity(h, want);
            if (cap > 0 && batch <= cap) break;
            want--;
        }
        const int rc = tmpc_set_latency_mode(h, want);
        static std::once_flag told;                      // (Solver instances solve concurrently from OpenMP threads, guidance_constraints.cpp:279: no plain static flag)
        if (rc == 1) {                                   // accepted, but this shape has no such variant: say so once instead of dropping the return value
            std::call_once(told, [want] { std::fprintf(stderr, "mpc_planner_solver (HIP): kernel variant %d is not available for this solver's shape; the library's fallback runs\n", want); });
        } else if (rc < 0) { std::fprintf(stderr, "tmpc_set_latency_mode: %s\n", tmpc_last_error(h)); std::exit(1); }
    }

    static std::string g_config_dir = "config";
    void setSolverConfigPath(const std::string &dir) { g_config_dir = dir; }

    static std::string trim(const std::string &s)
    {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    // flat "key: value" / "key: [a, b, c]" reader for the three generated YAML maps
    static void readFlatYaml(const std::string &path, std::vector<std::pair<std::string, std::vector<std::string>>> &out)
    {
        std::ifstream f(path);
        if (!f) { std::fprintf(stderr, "Solver: cannot open %s\n", path.c_str()); std::exit(1); }
        std::string line;
        while (std::getline(f, line)) {
            size_t c = line.find(':');
            if (c == std::string::npos) continue;
            std::string key = trim(line.substr(0, c)), val = trim(line.substr(c + 1));
            std::vector<std::string> items;
            if (!val.empty() && val[0] == '[') {
                std::stringstream ss(val.substr(1, val.size() - 2)); std::string it;
                while (std::getline(ss, it, ',')) items.push_back(trim(it));
            } else items.push_back(val);
            out.push_back({key, items});
        }
    }
    void loadSolverConfig(const std::string &dir, std::map<std::string, double> &settings, ParameterMap &pm, ModelMap &mm)
    {
        std::vector<std::pair<std::string, std::vector<std::string>>> kv;
        readFlatYaml(dir + "/solver_settings.yaml", kv);
        for (auto &e : kv) settings[e.first] = std::atof(e.second[0].c_str());
        kv.clear(); readFlatYaml(dir + "/parameter_map.yaml", kv);
        for (auto &e : kv) pm[e.first] = std::atoi(e.second[0].c_str());
        kv.clear(); readFlatYaml(dir + "/model_map.yaml", kv);
        for (auto &e : kv) mm[e.first] = ModelEntry{e.second[0], std::atoi(e.second[1].c_str()), std::atof(e.second[2].c_str()), std::atof(e.second[3].c_str())};
    }

    // ---- State (state.cpp:8-33) ----
    State::State()
    {
        std::map<std::string, double> cfg; ParameterMap pm;
        loadSolverConfig(g_config_dir, cfg, pm, _model_map);
        _nu = (int)cfg["nu"];
        _state = std::vector<double>((int)cfg["nx"], 0.0);
    }
    void State::initialize() { std::fill(_state.begin(), _state.end(), 0.0); }
    // The reference indexes _state[index - nu] for any name, i.e. out of bounds for an input (state.cpp:21-24; reached from
    // initializeWarmstart's k == 0 branch).  Inputs read as 0 here -- the same value tmpc_warmstart writes on the device.
    double State::get(std::string &&var_name) const
    {
        const ModelEntry &m = _model_map.at(var_name);
        return m.type == "x" ? _state[m.index - _nu] : 0.0;
    }
    Vector2d State::getPos() const { return Vector2d(get("x"), get("y")); }
    void State::set(std::string &&var_name, double value)
    {
        const ModelEntry &m = _model_map.at(var_name);
        if (m.type == "x") _state[m.index - _nu] = value;
    }

    AcadosParameters::AcadosParameters()
    {
        std::memset(xinit, 0, sizeof xinit); std::memset(x0, 0, sizeof x0); std::memset(all_parameters, 0, sizeof all_parameters);
    }
    Solver::AcadosOutput::AcadosOutput() { std::memset(xtraj, 0, sizeof xtraj); std::memset(utraj, 0, sizeof utraj); }
    void AcadosParameters::printParameters(const ParameterMap &parameter_map) const
    {
        std::printf("--- Parameters ---\n");
        for (int k = 0; k < SOLVER_N; k++) {
            std::printf("[%d]\n", k);
            for (auto &e : parameter_map)
                if (e.first != "num parameters") std::printf("  %s: %g\n", e.first.c_str(), all_parameters[k * SOLVER_NP + e.second]);
        }
    }
    void Solver::AcadosInfo::print() const
    {
        std::printf("--- Solver Info ---\nSQP iterations: %d\nMinimum time for solve [ms]: %g\nKKT: %g\nSolve Time [ms]: %g\nNLP Residuals: %g\n",
                    sqp_iter, min_time * 1000., kkt_norm_inf, solvetime * 1000., nlp_res);
    }
    void Solver::AcadosOutput::print() const
    {
        std::printf("\n--- xtraj ---\n");
        for (int k = 0; k <= SOLVER_N; k++) { for (int i = 0; i < NX; i++) std::printf("%14.6e", xtraj[k * NX + i]); std::printf("\n"); }
        std::printf("\n--- utraj ---\n");
        for (int k = 0; k < SOLVER_N; k++) { for (int i = 0; i < NU; i++) std::printf("%14.6e", utraj[k * NU + i]); std::printf("\n"); }
    }

    // ---- construction (acados_solver_interface.cpp:9-65) ----
    Solver::Solver(int solver_id)
    {
        _solver_id = solver_id;
        loadSolverConfig(g_config_dir, _config, _parameter_map, _model_map);
        N = SOLVER_N;
        nu = (unsigned)_config["nu"]; nx = (unsigned)_config["nx"]; nvar = (unsigned)_config["nvar"]; npar = (unsigned)_config["npar"];
        dt = _config["integrator_step"];
        _num_iterations = (int)_config["iterations"];
        if (nu != SOLVER_NU || nx != SOLVER_NX || npar != SOLVER_NP || (int)_config["N"] != SOLVER_N) {
            std::fprintf(stderr, "Solver: generated YAML maps do not match hip_solver_dims.h. Exiting.\n");
            std::exit(1);
        }
        reset();
    }
    Solver::~Solver()
    {
        const std::vector<BatchContext *> ctxs = _contexts;      // (forget() edits _contexts)
        for (BatchContext *c : ctxs) c->forget(this);
        if (_handle) tmpc_destroy(_handle);
    }

    // model_map.yaml -> tmpc_dims bounds.  The slack model's map has one more entry (`slack: [x, 7, 0, 5000]`,
    // solver_model.py:281-298) than tmpc_dims::lb/ub hold (TMPC_NV = 7): acados pins that state (DESIGN U9), it needs no bounds.
    template <typename Map>
    static void applyModelBounds(tmpc_dims &d, const Map &model_map)
    {
        for (auto &e : model_map) {
            const int i = e.second.index;
            if (i < 0 || i >= TMPC_NV) continue;
            d.lb[i] = e.second.lb; d.ub[i] = e.second.ub;
        }
    }

    void Solver::ensureHandle()
    {
        if (_handle) return;
        tmpc_dims d;
        tmpc_default_dims_ex(&d, SOLVER_N, SOLVER_S, SOLVER_NLIN, SOLVER_M, SOLVER_NSLK, SOLVER_SLACK);
        d.n_sqp = _num_iterations; d.dt = dt; d.cost_model = SOLVER_COST_MODEL; d.row_model = SOLVER_ROW_MODEL; d.npar = SOLVER_NP;
        applyModelBounds(d, _model_map);
        int status = tmpc_create(&_handle, &d, 1, _device);
        if (status) {                                   // reference: exit(1) when the capsule cannot be created (:35-39)
            std::printf("tmpc_create() returned status %d (no MI355X / library not built). Exiting.\n", status);
            std::exit(1);
        }
        applyTickVariant(_handle, 1);                   // a Solver serves control ticks of a few planners: latency variant
        tmpc_enable_timing(_handle, 4);                 // HIP events around every launch -> _info.elapsed_time / solvetime / min_time
    }

    Solver &Solver::operator=(const Solver &rhs) { _params = rhs._params; return *this; }      // (:67-77)
    void Solver::reset() { _params = AcadosParameters(); _info = AcadosInfo(); _output = AcadosOutput(); }  // (:79-84)

    // ---- solve (:86-204).  The reference's wall-clock early exit (:111-116) is intentionally not reproduced (non-deterministic):
    // the RTI loop has the fixed budget _num_iterations and runs in ONE launch; it is bitwise the same as the one-iteration
    // protocol below (tests/test_gpu_iterations.py, tests/cpp/test_solver.cpp). ----
    int Solver::iterationBudget() const
    {
        if (_iteration_time_estimate > 0. && _params.solver_timeout > 0.) {
            const int n = (int)std::floor(_params.solver_timeout / _iteration_time_estimate);
            return std::min(_num_iterations, std::max(1, n));
        }
        return _num_iterations;
    }
    int Solver::solve()
    {
        initializeOneIteration();
        runIterations(iterationBudget(), true);
        _iterations_done = 0;                                   // (sqp_iter of the call is the whole loop's count)
        return completeOneIteration();
    }
    void Solver::initializeOneIteration()
    {
        ensureHandle();
        _info = AcadosInfo();
        _iterations_done = 0;
        _new_solve = true;
        if (tmpc_set_batch(_handle, 1, _params.xinit, _params.x0, _params.all_parameters)) {
            std::fprintf(stderr, "tmpc_set_batch: %s\n", tmpc_last_error(_handle)); std::exit(1);
        }
    }
    // n RTI iterations from the capsule's state: multipliers always kept (the reference never clears them except after a failed
    // solve), primal iterate kept unless loadWarmstart() asked for _params.x0 (:274-284)
    int Solver::runIterations(int n, bool complete)
    {
        // the first iteration call after initializeOneIteration() opens a new solve: a loop exit of the previous solve (:105-106) does not carry over
        const int flags = TMPC_ITER_KEEP_MULTIPLIERS | (_warmstart_pending ? 0 : TMPC_ITER_KEEP_ITERATE) | (complete ? TMPC_ITER_COMPLETE : 0) |
                          (_new_solve ? TMPC_ITER_NEW_SOLVE : 0);
        if (tmpc_solve_iterations(_handle, n, flags)) { std::fprintf(stderr, "tmpc_solve_iterations: %s\n", tmpc_last_error(_handle)); std::exit(1); }
        _new_solve = false;
        float ms[4] = {0.f, 0.f, 0.f, 0.f}; int32_t n_ms = 0;   // (:151-153) the launch's time, HIP events on the handle's stream (drained every call)
        if (tmpc_get_timings(_handle, ms, 4, &n_ms) == 0 && n_ms > 0 && n > 0) recordTiming(ms[n_ms - 1] * 1e-3, n);
        if (n > 0) _warmstart_pending = false;
        return 0;
    }
    int Solver::solveOneIteration()
    {
        if (_iterations_done > 0 && _info.qp_status != 0) return _exit_code_one_iter == 1 ? 0 : 4;   // the loop has ended (:105-106)
        runIterations(1, false);
        int32_t exit_code = 0, qp_status = 0;
        if (tmpc_get(_handle, nullptr, nullptr, nullptr, &exit_code, &qp_status, nullptr, nullptr, nullptr)) {
            std::fprintf(stderr, "tmpc_get: %s\n", tmpc_last_error(_handle)); std::exit(1);
        }
        _iterations_done++;
        _info.qp_status = qp_status;
        _exit_code_one_iter = exit_code;
        return (qp_status == 0 || qp_status == 2) ? 0 : 4;      // ACADOS_SUCCESS / ACADOS_QP_FAILURE (DESIGN U5)
    }
    int Solver::completeOneIteration()
    {
        const int done = _iterations_done, last_qp = _info.qp_status;
        if (done > 0) runIterations(0, true);                   // completeOneIteration of the one-iteration protocol: evaluation + reset on failure
        int32_t exit_code = 0, qp_status = 0, sqp_iter = 0, qp_it = 0; double res_eq = 0.;
        if (tmpc_get(_handle, _output.xtraj, _output.utraj, &_info.pobj, &exit_code, &qp_status, &sqp_iter, &res_eq, &qp_it)) {
            std::fprintf(stderr, "tmpc_get: %s\n", tmpc_last_error(_handle)); std::exit(1);
        }
        _info.qp_status = done > 0 ? last_qp : qp_status; _info.sqp_iter = done > 0 ? done : sqp_iter; _info.nlp_res = res_eq;
        _exit_code_one_iter = exit_code;
        return _exit_code_one_iter;
    }

    std::vector<int> Solver::solveBatch(BatchContext &ctx, const std::vector<Solver *> &solvers) { return ctx.solve(solvers); }

    // (:151-153) time_tot of the last Solver_acados_solve -> elapsed_time, accumulated in solvetime, minimum in min_time.  Here the
    // time is that of the HIP launch that did `iterations` RTI iterations (HIP events on the handle's stream, tmpc_get_timings).
    void Solver::recordTiming(double launch_seconds, int iterations)
    {
        const double per_iteration = launch_seconds / (iterations > 0 ? iterations : 1);
        _info.elapsed_time = per_iteration;
        _info.solvetime += launch_seconds;
        _info.min_time = std::min(per_iteration, _info.min_time);
    }

    BatchContext::~BatchContext()
    {
        for (auto &e : _slot) {                                  // the Solvers outlive the context: they must not call back into it
            auto &cs = const_cast<Solver *>(e.first)->_contexts;
            cs.erase(std::remove(cs.begin(), cs.end(), this), cs.end());
        }
        if (_handle) tmpc_destroy(_handle);
    }
    // A forgotten Solver's slot goes back to the pool; it is cleared (tmpc_clear_slot: the slot's next launch starts like a new capsule)
    // before it is handed to another Solver, so neither a new Solver at the same address nor the next owner inherits multipliers.
    void BatchContext::forget(const Solver *s)
    {
        auto it = _slot.find(s);
        if (it == _slot.end()) return;
        _free_slots.push_back(it->second);
        _slot.erase(it);
        auto &cs = const_cast<Solver *>(s)->_contexts;
        cs.erase(std::remove(cs.begin(), cs.end(), this), cs.end());
    }
    int BatchContext::takeSlot()
    {
        if (_free_slots.empty()) return _next_slot++;
        const int slot = _free_slots.back(); _free_slots.pop_back();
        if (_handle && tmpc_clear_slot(_handle, slot)) { std::fprintf(stderr, "tmpc_clear_slot: %s\n", tmpc_last_error(_handle)); std::exit(1); }
        return slot;
    }

    // (re)create the handle when the solver settings change, grow it (keeping every slot's state) when more Solvers appear than it holds
    void BatchContext::ensure(const Solver *s0, int needed_slots)
    {
        const bool settings_changed = _handle && (_device != s0->_device || _iterations != s0->_num_iterations || _dt != s0->dt);
        if (_handle && !settings_changed && needed_slots <= _capacity) return;
        const int cap = settings_changed ? std::max(_capacity, needed_slots) : std::max(8, 2 * needed_slots);
        tmpc_dims d; tmpc_default_dims_ex(&d, SOLVER_N, SOLVER_S, SOLVER_NLIN, SOLVER_M, SOLVER_NSLK, SOLVER_SLACK);
        d.n_sqp = s0->_num_iterations; d.dt = s0->dt; d.cost_model = SOLVER_COST_MODEL; d.row_model = SOLVER_ROW_MODEL; d.npar = SOLVER_NP;
        applyModelBounds(d, s0->_model_map);
        tmpc_handle *h = nullptr;
        if (tmpc_create(&h, &d, cap, s0->_device)) { std::printf("tmpc_create() failed. Exiting.\n"); std::exit(1); }
        tmpc_enable_timing(h, 4);
        if (_handle) {
            if (!settings_changed && tmpc_copy_state(h, _handle)) { std::fprintf(stderr, "tmpc_copy_state: %s\n", tmpc_last_error(h)); std::exit(1); }
            if (settings_changed) {                                       // other iteration budget / dt / device: a different solver, nothing to carry over
                for (auto &e : _slot) { auto &cs = const_cast<Solver *>(e.first)->_contexts; cs.erase(std::remove(cs.begin(), cs.end(), this), cs.end()); }
                _slot.clear(); _free_slots.clear(); _next_slot = 0;
            }
            tmpc_destroy(_handle);
        }
        _handle = h; _capacity = cap; _device = s0->_device; _iterations = s0->_num_iterations; _dt = s0->dt;
    }

    std::vector<int> BatchContext::solve(const std::vector<Solver *> &solvers_in)
    {
        std::vector<int> codes_in(solvers_in.size(), 0);
        if (solvers_in.empty()) return codes_in;
        // a Solver named twice is solved once (one capsule, one slot); every mention gets its exit code
        std::vector<Solver *> solvers;
        std::vector<int> first_of(solvers_in.size());
        {
            std::map<const Solver *, int> seen;
            for (size_t i = 0; i < solvers_in.size(); i++) {
                auto it = seen.find(solvers_in[i]);
                if (it == seen.end()) { seen[solvers_in[i]] = (int)solvers.size(); first_of[i] = (int)solvers.size(); solvers.push_back(solvers_in[i]); }
                else first_of[i] = it->second;
            }
        }
        const int B = (int)solvers.size();
        std::vector<int> codes(B, 0);
        Solver *s0 = solvers[0];
        int fresh = 0;
        for (Solver *s : solvers) if (!_slot.count(s)) fresh++;
        const int from_pool = std::min(fresh, (int)_free_slots.size());
        ensure(s0, _next_slot + fresh - from_pool);
        for (Solver *s : solvers)
            if (!_slot.count(s)) { _slot[s] = takeSlot(); s->_contexts.push_back(this); }   // (a reused slot was cleared: the previous owner's state is not inherited)
        const size_t n0 = (SOLVER_NU + SOLVER_NX) * (SOLVER_N + 1), np = (size_t)SOLVER_NP * SOLVER_N;
        // (members, grown as needed: a tick does not allocate and zero ~23 KB per planner)
        std::vector<double> &xinit = _stage_xinit, &x0 = _stage_x0, &par = _stage_par;
        if (xinit.size() < (size_t)B * SOLVER_NX) { xinit.resize((size_t)B * SOLVER_NX); x0.resize(B * n0); par.resize(B * np); }
        std::vector<int32_t> slots(B);
        for (int b = 0; b < B; b++) {
            std::memcpy(&xinit[(size_t)b * SOLVER_NX], solvers[b]->_params.xinit, sizeof(double) * SOLVER_NX);
            std::memcpy(&x0[b * n0], solvers[b]->_params.x0, sizeof(double) * n0);
            std::memcpy(&par[b * np], solvers[b]->_params.all_parameters, sizeof(double) * np);
            slots[b] = _slot.at(solvers[b]);
        }
        const size_t nxt = (size_t)SOLVER_NX * (SOLVER_N + 1), nut = (size_t)SOLVER_NU * SOLVER_N;
        std::vector<double> xt(B * nxt), ut(B * nut), pobj(B), res(B);
        std::vector<int32_t> ec(B), qs(B), si(B), qi(B);
        // entry b = solvers[b] on ITS slot: like the planners' own capsules, every Solver keeps its multipliers from tick to tick (and
        // loses them after a failed solve); GuidanceConstraints::optimize loads every planner's warm start (:337), so the iterate comes
        // from x0.  A new solve() of every Solver: loop exits of the previous tick do not carry over.
        float ms[4] = {0.f, 0.f, 0.f, 0.f}; int32_t n_ms = 0;
        applyTickVariant(_handle, B);                   // same variant as solve() while the batch is a tick (bitwise equal to it); throughput kernels above
        if (tmpc_set_batch(_handle, B, xinit.data(), x0.data(), par.data()) || tmpc_set_slots(_handle, slots.data()) ||
            tmpc_solve_iterations(_handle, s0->_num_iterations, TMPC_ITER_KEEP_MULTIPLIERS | TMPC_ITER_COMPLETE | TMPC_ITER_NEW_SOLVE) ||
            tmpc_get(_handle, xt.data(), ut.data(), pobj.data(), ec.data(), qs.data(), si.data(), res.data(), qi.data()) ||
            tmpc_get_timings(_handle, ms, 4, &n_ms)) {
            std::fprintf(stderr, "solveBatch: %s\n", tmpc_last_error(_handle)); std::exit(1);
        }
        _last_launch_s = n_ms > 0 ? ms[n_ms - 1] * 1e-3 : 0.;
        for (int b = 0; b < B; b++) {
            Solver *s = solvers[b];
            std::memcpy(s->_output.xtraj, &xt[b * nxt], sizeof(double) * nxt);
            std::memcpy(s->_output.utraj, &ut[b * nut], sizeof(double) * nut);
            s->_info = Solver::AcadosInfo(); s->_info.pobj = pobj[b]; s->_info.qp_status = qs[b]; s->_info.sqp_iter = si[b]; s->_info.nlp_res = res[b];
            s->recordTiming(_last_launch_s, s0->_num_iterations);      // (the batch is one launch: every Solver of it sees that launch's time)
            s->_exit_code_one_iter = ec[b]; codes[b] = ec[b];
        }
        for (size_t i = 0; i < solvers_in.size(); i++) codes_in[i] = codes[first_of[i]];
        return codes_in;
    }

    // ---- parameters / xinit / warm start / output: host-only, identical index arithmetic (:206-389) ----
    bool Solver::hasParameter(std::string &&parameter) { return _parameter_map.count(parameter) > 0; }
    void Solver::setParameter(int k, std::string &&parameter, double value) { _params.all_parameters[k * npar + _parameter_map.at(parameter)] = value; }
    void Solver::setParameter(int k, std::string &parameter, double value) { _params.all_parameters[k * npar + _parameter_map.at(parameter)] = value; }
    double Solver::getParameter(int k, std::string &&parameter) { return _params.all_parameters[k * npar + _parameter_map.at(parameter)]; }
    void Solver::setXinit(std::string &&state_name, double value) { _params.xinit[_model_map.at(state_name).index - nu] = value; }
    void Solver::setXinit(const State &state)
    {
        for (auto &e : _model_map)
            if (e.second.type == "x") setXinit(std::string(e.first), state.get(std::string(e.first)));
    }
    void Solver::setEgoPrediction(unsigned int k, std::string &&var_name, double value) { _params.x0[k * nvar + _model_map.at(var_name).index] = value; }
    double Solver::getEgoPrediction(unsigned int k, std::string &&var_name) { return _params.x0[k * nvar + _model_map.at(var_name).index]; }
    void Solver::setEgoPredictionPosition(unsigned int k, const Vector2d &value) { setEgoPrediction(k, "x", value(0)); setEgoPrediction(k, "y", value(1)); }
    Vector2d Solver::getEgoPredictionPosition(unsigned int k) { return Vector2d(getEgoPrediction(k, "x"), getEgoPrediction(k, "y")); }
    // (:274-284) ocp_nlp_out_set(x, u) from _params.x0: the next iteration starts from the warm start; the multipliers stay
    void Solver::loadWarmstart() { _warmstart_pending = true; }

    void Solver::initializeWithState(const State &initial_state)          // (:286-301)
    {
        for (int k = 0; k <= N; k++)
            for (auto &e : _model_map)
                setEgoPrediction(k, std::string(e.first), e.second.type == "x" ? initial_state.get(std::string(e.first)) : 0.);
    }
    void Solver::initializeWithBraking(const State &initial_state)        // (:303-342); deceleration_at_infeasible = 3.0
    {
        initializeWithState(initial_state);
        double x = initial_state.get("x"), y = initial_state.get("y"), psi = initial_state.get("psi"), v = initial_state.get("v");
        double spline = initial_state.get("spline");
        const double deceleration = _config.count("deceleration_at_infeasible") ? std::fabs(_config["deceleration_at_infeasible"]) : 3.0;
        const double a = -deceleration;
        for (int k = 0; k <= N; k++) {
            if (k > 0) { x += v * dt * std::cos(psi); y += v * dt * std::sin(psi); spline += v * dt; v += a * dt; v = std::max(v, 0.); }
            setEgoPrediction(k, "x", x); setEgoPrediction(k, "y", y); setEgoPrediction(k, "psi", psi); setEgoPrediction(k, "v", v);
            setEgoPrediction(k, "spline", spline); setEgoPrediction(k, "a", a); setEgoPrediction(k, "w", 0);
        }
    }
    void Solver::initializeWarmstart(const State &initial_state, bool shift_previous_solution_forward)   // (:344-376)
    {
        if (_stage_indexing == StageIndexing::ForcesStages) {
            // forces_solver_interface.cpp:147-182: N stages; shifted: [state, x_2, ..., x_{N-1}, x_{N-1}], kept: [state, x_1, ..., x_{N-1}];
            // this solver's node N (not a Forces stage) repeats the terminal stage
            for (int k = 0; k < N; k++)
                for (auto &e : _model_map) {
                    std::string n = e.first;
                    if (k == 0) setEgoPrediction(0, std::string(n), initial_state.get(std::string(n)));
                    else if (!shift_previous_solution_forward || k == N - 1) setEgoPrediction(k, std::string(n), getOutput(k, std::string(n)));
                    else setEgoPrediction(k, std::string(n), getOutput(k + 1, std::string(n)));
                }
            for (auto &e : _model_map) setEgoPrediction(N, std::string(e.first), getOutput(N - 1, std::string(e.first)));
            return;
        }
        if (shift_previous_solution_forward) {
            for (int k = 0; k <= N; k++)
                for (auto &e : _model_map) {
                    std::string n = e.first;
                    if (k == 0) setEgoPrediction(0, std::string(n), initial_state.get(std::string(n)));
                    else if (k == N - 1) setEgoPrediction(N - 1, std::string(n), getOutput(N - 1, std::string(n)));
                    else if (k == N) setEgoPrediction(N, std::string(n), getOutput(N - 1, std::string(n)));
                    else setEgoPrediction(k, std::string(n), getOutput(k + 1, std::string(n)));
                }
        } else {
            for (int k = 0; k < N; k++)
                for (auto &e : _model_map) setEgoPrediction(k, std::string(e.first), getOutput(k, std::string(e.first)));
        }
    }
    double Solver::getOutput(int k, std::string &&state_name) const      // (:379-389); Forces mode: forces_solver_interface.cpp:241-244
    {
        if (_stage_indexing == StageIndexing::ForcesStages && (k < 0 || k >= N)) {
            std::fprintf(stderr, "Solver::getOutput: stage %d requested, the Forces indexing has stages 0 .. %d\n", k, N - 1);
            std::abort();
        }
        const ModelEntry &m = _model_map.at(state_name);
        return m.type == "x" ? _output.xtraj[k * nx + m.index - nu] : _output.utraj[k * nu + m.index];
    }
    std::string Solver::explainExitFlag(int exitflag) const              // (:391-424)
    {
        switch (exitflag) {
        case 1: return "Success";
        case 0: return "Failure (no more information)";
        case 2: return "Failure (maximum number of iterations reached)";
        case 3: return "Failure (minimum step size reached)";
        case 4: break;
        default: return "Unknown exit code";
        }
        switch (_info.qp_status) {
        case 1: return "QP Failure: No more information on QP failure";
        case 2: return "QP Failure: Max Iterations";
        case 3: return "QP Failure: Minimal Step Reached";
        case 4: return "QP Failure: NAN in solution";
        case 5: return "QP Failure: Inconsistent Equality Constraints";
        default: return "QP Failure: UNKNOWN";
        }
    }
    void Solver::printIfBoundLimited() const                             // (:426-446)
    {
        for (int k = 0; k < N; k++)
            for (auto &e : _model_map) {
                if (k == 0 && e.second.type == "x") continue;
                const double val = getOutput(k, std::string(e.first));
                if (std::fabs(val - e.second.lb) < 1e-2) std::printf("%s limited by lower bound\n", e.first.c_str());
                if (std::fabs(val - e.second.ub) < 1e-2) std::printf("%s limited by upper bound\n", e.first.c_str());
            }
    }
}
