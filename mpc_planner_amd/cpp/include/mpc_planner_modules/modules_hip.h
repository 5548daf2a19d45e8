/*
 * mpc_planner_modules/modules_hip.h -- C++ host side of the accelerated path, restated after
 * the reference's modules so that the batched optimize() of INTEGRATION.md section 4 exists as compiled, tested code:
 *
 *   MPCBaseModule::setParameters          mpc_planner_modules/src/mpc_base.cpp:23-35
 *   Contouring::setParameters             mpc_planner_modules/src/contouring.cpp:50-124
 *   EllipsoidConstraints::update / setParameters      ellipsoid_constraints.cpp:23-90
 *   LinearizedConstraints::update / projectToSafety / setParameters      linearized_constraints.cpp:49-189
 *   GuidanceConstraints::optimize / initializeSolverWithGuidance / FindBestPlanner      guidance_constraints.cpp:264-434
 *   ScenarioConstraints::optimize         scenario_constraints.cpp:58-108
 *
 * Same class and method names, same member semantics; what the reference takes from packages that are not in its tree
 * is an explicit input here: CONFIG[...] (mpc_planner_util, yaml-cpp) -> ModuleConfig; guidance_planner::GlobalGuidance ->
 * std::vector<GuidanceTrajectory>; RosTools::Spline2D -> ModuleData::path segments; scenario_module's sampler / polygon
 * construction -> halfspaces handed in per scenario solver.  The OpenMP loop over local planners becomes: prepare every planner's
 * parameters on the host (same statements, same order), ONE Solver::solveBatch launch, then the reference's bookkeeping.
 * Header-only: everything is small and is compiled against the generated setSolverParameter* functions.
 *
 * STATUS: integration scaffolding, not product.  This file exists so that the patch of INTEGRATION.md section 4 compiles and can be tested
 * end to end without the reference tree; a maintainer applies the patch to the reference's own modules instead of taking this file.
 * The product is the C-ABI library (include/tmpc_hip.h, csrc/) and the Solver / BatchContext classes of solver_interface.h; the
 * module bodies below restate reference statements by necessity (same parameter names, same order of writes).
 */
#ifndef MPC_PLANNER_MODULES_HIP_H
#define MPC_PLANNER_MODULES_HIP_H

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <mpc_planner_solver/mpc_planner_parameters.h>
#include <mpc_planner_solver/solver_interface.h>
#include <mpc_planner_types/data_types.h>
#include <mpc_planner_types/module_data.h>
#include <mpc_planner_types/realtime_data.h>

namespace MPCPlanner
{
    /* The CONFIG[...] entries the path reads (settings.yaml, guidance_planner.yaml). */
    struct ModuleConfig
    {
        int n_discs{1};
        double robot_radius{0.325};
        double risk{0.05};                                   /* probabilistic/risk */
        double obstacle_radius{0.35};                        /* obstacle_radius (GaussianConstraints uses the configured value, not the obstacle's own) */
        int max_obstacles{8};
        int n_other_halfspaces{0};                           /* linearized_constraints/add_halfspaces */
        std::map<std::string, double> weights;               /* weights/<name> */
        bool dynamic_velocity_reference{false};              /* contouring/dynamic_velocity_reference */
        int num_segments{5};                                 /* contouring/num_segments */
        bool use_tmpcpp{true};                               /* t-mpc/use_t-mpc++ */
        bool enable_constraints{true};                       /* t-mpc/enable_constraints */
        bool warmstart_with_mpc_solution{false};             /* t-mpc/warmstart_with_mpc_solution */
        bool shift_previous_solution_forward{false};
        int n_paths{4};                                      /* guidance_planner.yaml n_paths */
        double selection_weight_consistency{0.75};
    };

    inline double ExponentialQuantile(double lambda, double p) { return -std::log(1. - p) / lambda; }     /* ros_tools/math [UPSTREAM formula] */

    /* ---- mpc_base.cpp:23-35 ---- */
    class MPCBaseModule
    {
    public:
        MPCBaseModule(std::shared_ptr<Solver> solver, const ModuleConfig &cfg, const std::vector<std::string> &weight_names)
            : _solver(solver), _cfg(cfg), _weight_names(weight_names) {}
        void setParameters(const RealTimeData &, const ModuleData &, int k)
        {
            for (auto &weight : _weight_names) { std::string w = weight; _solver->setParameter(k, w, _cfg.weights.at(weight)); }
        }
        std::shared_ptr<Solver> _solver;
    private:
        ModuleConfig _cfg;
        std::vector<std::string> _weight_names;
    };

    /* ---- contouring.cpp:50-124 ---- */
    class Contouring
    {
    public:
        Contouring(std::shared_ptr<Solver> solver, const ModuleConfig &cfg) : _solver(solver), _cfg(cfg) {}
        void update(State &, const RealTimeData &, ModuleData &module_data) { _segments = module_data.path; }
        void setParameters(const RealTimeData &, const ModuleData &, int k)
        {
            setSolverParameterContour(k, _solver->_params, _cfg.weights.at("contour"));
            setSolverParameterLag(k, _solver->_params, _cfg.weights.at("lag"));
            setSolverParameterTerminalAngle(k, _solver->_params, _cfg.weights.at("terminal_angle"));
            setSolverParameterTerminalContouring(k, _solver->_params, _cfg.weights.at("terminal_contouring"));
            for (int i = 0; i < _cfg.num_segments; i++) {                       /* setSplineParameters (:94-124) */
                const PathSegment &sg = _segments.at(i);
                setSolverParameterSplineXA(k, _solver->_params, sg.ax, i); setSolverParameterSplineXB(k, _solver->_params, sg.bx, i);
                setSolverParameterSplineXC(k, _solver->_params, sg.cx, i); setSolverParameterSplineXD(k, _solver->_params, sg.dx, i);
                setSolverParameterSplineYA(k, _solver->_params, sg.ay, i); setSolverParameterSplineYB(k, _solver->_params, sg.by, i);
                setSolverParameterSplineYC(k, _solver->_params, sg.cy, i); setSolverParameterSplineYD(k, _solver->_params, sg.dy, i);
                setSolverParameterSplineStart(k, _solver->_params, sg.start, i);
            }
        }
        std::shared_ptr<Solver> _solver;
    private:
        ModuleConfig _cfg;
        std::vector<PathSegment> _segments;
    };

#ifndef SOLVER_ROW_MODEL          /* (dims headers written before the Gaussian rows existed) */
#define SOLVER_ROW_MODEL 0
#endif

#if SOLVER_M > 0 && SOLVER_ROW_MODEL == 1
    /* ---- gaussian_constraints.cpp:22-79 (the solver was generated with gaussian=True: tmpc_dims::row_model = 1) ---- */
    class GaussianConstraints
    {
    public:
        GaussianConstraints(std::shared_ptr<Solver> solver, const ModuleConfig &cfg) : _solver(solver), _cfg(cfg) {}
        void update(State &state, const RealTimeData &, ModuleData &)
        {
            _dummy_x = state.get("x") + 100.; _dummy_y = state.get("y") + 100.;
        }
        void setParameters(const RealTimeData &data, const ModuleData &, int k)
        {
            setSolverParameterEgoDiscRadius(k, _solver->_params, _cfg.robot_radius);
            for (int d = 0; d < _cfg.n_discs; d++) setSolverParameterEgoDiscOffset(k, _solver->_params, data.robot_area[d].offset, d);
            if (k == 0) {                                                        /* dummies (:41-54) */
                for (size_t i = 0; i < data.dynamic_obstacles.size(); i++) {
                    setSolverParameterGaussianObstX(k, _solver->_params, _dummy_x, i); setSolverParameterGaussianObstY(k, _solver->_params, _dummy_y, i);
                    setSolverParameterGaussianObstMajor(k, _solver->_params, 0.1, i); setSolverParameterGaussianObstMinor(k, _solver->_params, 0.1, i);
                    setSolverParameterGaussianObstRisk(k, _solver->_params, 0.05, i); setSolverParameterGaussianObstR(k, _solver->_params, 0.1, i);
                }
                return;
            }
            for (size_t i = 0; i < data.dynamic_obstacles.size(); i++) {
                const auto &obstacle = data.dynamic_obstacles[i];
                if (obstacle.prediction.type != PredictionType::GAUSSIAN) continue;                  /* (:62) */
                const auto &mode = obstacle.prediction.modes[0];
                setSolverParameterGaussianObstX(k, _solver->_params, mode[k - 1].position(0), i);
                setSolverParameterGaussianObstY(k, _solver->_params, mode[k - 1].position(1), i);
                const bool dynamic = obstacle.type == ObstacleType::DYNAMIC;                        /* static obstacles have no uncertainty (:72-76) */
                setSolverParameterGaussianObstMajor(k, _solver->_params, dynamic ? mode[k - 1].major_radius : 0.001, i);
                setSolverParameterGaussianObstMinor(k, _solver->_params, dynamic ? mode[k - 1].minor_radius : 0.001, i);
                setSolverParameterGaussianObstRisk(k, _solver->_params, _cfg.risk, i);
                setSolverParameterGaussianObstR(k, _solver->_params, _cfg.obstacle_radius, i);
            }
        }
        bool isDataReady(const RealTimeData &data, std::string &missing_data) const                  /* :81-105 */
        {
            if ((int)data.dynamic_obstacles.size() != _cfg.max_obstacles) { missing_data += "Obstacles "; return false; }
            for (const auto &o : data.dynamic_obstacles) {
                if (o.prediction.modes.empty()) { missing_data += "Obstacle Prediction "; return false; }
                if (o.prediction.type != PredictionType::GAUSSIAN) { missing_data += "Obstacle Prediction (Type is not Gaussian) "; return false; }
            }
            return true;
        }
        std::shared_ptr<Solver> _solver;
    private:
        ModuleConfig _cfg;
        double _dummy_x{0.}, _dummy_y{0.};
    };
#define GUIDANCE_CONSTRAINTS_TYPE GaussianConstraints      /* what the reference's generator writes into modules.h (guidance_constraints.py:56-62) */
#endif

#if SOLVER_M > 0 && SOLVER_ROW_MODEL == 0
#define GUIDANCE_CONSTRAINTS_TYPE EllipsoidConstraints
    /* ---- ellipsoid_constraints.cpp:23-90 ---- */
    class EllipsoidConstraints
    {
    public:
        EllipsoidConstraints(std::shared_ptr<Solver> solver, const ModuleConfig &cfg) : _solver(solver), _cfg(cfg) {}
        void update(State &state, const RealTimeData &, ModuleData &)
        {
            _dummy_x = state.get("x") + 50; _dummy_y = state.get("y") + 50;
        }
        void setParameters(const RealTimeData &data, const ModuleData &, int k)
        {
            setSolverParameterEgoDiscRadius(k, _solver->_params, _cfg.robot_radius);
            for (int d = 0; d < _cfg.n_discs; d++) setSolverParameterEgoDiscOffset(k, _solver->_params, data.robot_area[d].offset, d);
            if (k == 0) {                                                        /* dummies (:42-56) */
                for (size_t i = 0; i < data.dynamic_obstacles.size(); i++) {
                    setSolverParameterEllipsoidObstX(0, _solver->_params, _dummy_x, i); setSolverParameterEllipsoidObstY(0, _solver->_params, _dummy_y, i);
                    setSolverParameterEllipsoidObstPsi(0, _solver->_params, 0., i); setSolverParameterEllipsoidObstR(0, _solver->_params, 0.1, i);
                    setSolverParameterEllipsoidObstMajor(0, _solver->_params, 0., i); setSolverParameterEllipsoidObstMinor(0, _solver->_params, 0., i);
                    setSolverParameterEllipsoidObstChi(0, _solver->_params, 1., i);
                }
                return;
            }
            for (size_t i = 0; i < data.dynamic_obstacles.size(); i++) {
                const auto &obstacle = data.dynamic_obstacles[i];
                const auto &mode = obstacle.prediction.modes[0];
                /* the first prediction step is index 1 of the optimisation problem: k-1 maps to the predictions for this stage */
                setSolverParameterEllipsoidObstX(k, _solver->_params, mode[k - 1].position(0), i);
                setSolverParameterEllipsoidObstY(k, _solver->_params, mode[k - 1].position(1), i);
                setSolverParameterEllipsoidObstPsi(k, _solver->_params, mode[k - 1].angle, i);
                setSolverParameterEllipsoidObstR(k, _solver->_params, obstacle.radius, i);
                if (obstacle.prediction.type == PredictionType::DETERMINISTIC) {
                    setSolverParameterEllipsoidObstMajor(k, _solver->_params, 0., i); setSolverParameterEllipsoidObstMinor(k, _solver->_params, 0., i);
                    setSolverParameterEllipsoidObstChi(k, _solver->_params, 1., i);
                } else if (obstacle.prediction.type == PredictionType::GAUSSIAN) {
                    const double chi = ExponentialQuantile(0.5, 1.0 - _cfg.risk);
                    setSolverParameterEllipsoidObstMajor(k, _solver->_params, mode[k - 1].major_radius, i);
                    setSolverParameterEllipsoidObstMinor(k, _solver->_params, mode[k - 1].minor_radius, i);
                    setSolverParameterEllipsoidObstChi(k, _solver->_params, chi, i);
                }
            }
        }
        std::shared_ptr<Solver> _solver;
    private:
        ModuleConfig _cfg;
        double _dummy_x{0.}, _dummy_y{0.};
    };
#endif

#if SOLVER_NLIN > 0
    /* Douglas-Rachford step onto the outside of two discs of radius r (ros_tools' DouglasRachford is not in the reference tree;
     * restated from the published operator x <- (x + R_A R_B x) / 2 with reflections R = 2 P - I, P = nearest point outside the
     * disc, in the reference's call order: anchor first, then the obstacle; applied only when p collides with `delta`).  Same
     * arithmetic as mpc_planner_amd/modules.py::project_to_safety and tmpc_linearize_topology_kernel. */
    struct DouglasRachford
    {
        static void outside(double px, double py, double cx, double cy, double r, double &ox, double &oy)
        {
            const double dx = px - cx, dy = py - cy;
            const double dist = std::sqrt(dx * dx + dy * dy);
            if (dist >= r) { ox = px; oy = py; }
            else if (dist > 1e-12) { const double s = r / dist; ox = cx + dx * s; oy = cy + dy * s; }
            else { ox = cx; oy = cy + r; }
        }
        void douglasRachfordProjection(Vector2d &p, const Vector2d &delta, const Vector2d &anchor, const double r, const Vector2d &) const
        {
            const double dx = p(0) - delta(0), dy = p(1) - delta(1);
            if (std::sqrt(dx * dx + dy * dy) >= r) return;
            double ax, ay, bx, by;
            outside(p(0), p(1), anchor(0), anchor(1), r, ax, ay);
            const double rx = 2. * ax - p(0), ry = 2. * ay - p(1);                 /* reflect at the anchor's set */
            outside(rx, ry, delta(0), delta(1), r, bx, by);
            const double sx = 2. * bx - rx, sy = 2. * by - ry;                     /* reflect at the obstacle's set */
            p = Vector2d((p(0) + sx) / 2., (p(1) + sy) / 2.);
        }
    };

    /* ---- linearized_constraints.cpp:49-189 (topology mode: one disc, the robot position; :43-47) ---- */
    class LinearizedConstraints
    {
    public:
        LinearizedConstraints(std::shared_ptr<Solver> solver, const ModuleConfig &cfg)
            : _solver(solver), _cfg(cfg), _max_obstacles(cfg.max_obstacles), _n_other_halfspaces(cfg.n_other_halfspaces)
        {
            const int n = _max_obstacles + _n_other_halfspaces;
            _a1.assign(1, std::vector<std::vector<double>>(solver->N, std::vector<double>(n, 0.)));
            _a2 = _a1; _b = _a1;
        }
        void setTopologyConstraints() { _n_discs = 1; _use_guidance = true; }
        void update(State &state, const RealTimeData &data, ModuleData &module_data)
        {
            _dummy_b = state.get("x") + 100.;
            std::vector<DynamicObstacle> copied_obstacles = data.dynamic_obstacles;
            _num_obstacles = (int)copied_obstacles.size();
            for (int k = 1; k < _solver->N; k++) {
                for (int d = 0; d < _n_discs; d++) {
                    Vector2d pos(_solver->getEgoPrediction(k, "x"), _solver->getEgoPrediction(k, "y"));
                    if (!_use_guidance) {
                        auto &disc = data.robot_area[d];
                        Vector2d disc_pos = disc.getPosition(pos, _solver->getEgoPrediction(k, "psi"));
                        projectToSafety(copied_obstacles, k, disc_pos);
                        pos = disc_pos;
                    } else {
                        projectToSafety(copied_obstacles, k, pos);
                    }
                    for (size_t obs_id = 0; obs_id < copied_obstacles.size(); obs_id++) {
                        const auto &copied_obstacle = copied_obstacles[obs_id];
                        const Vector2d &obstacle_pos = copied_obstacle.prediction.modes[0][k - 1].position;
                        const double diff_x = obstacle_pos(0) - pos(0), diff_y = obstacle_pos(1) - pos(1);
                        const double dist = std::sqrt(diff_x * diff_x + diff_y * diff_y);
                        _a1[d][k][obs_id] = diff_x / dist;
                        _a2[d][k][obs_id] = diff_y / dist;
                        const double radius = _use_guidance ? 1e-3 : copied_obstacle.radius;
                        _b[d][k][obs_id] = _a1[d][k][obs_id] * obstacle_pos(0) + _a2[d][k][obs_id] * obstacle_pos(1) - (radius + _cfg.robot_radius);
                    }
                    if (!module_data.static_obstacles.empty()) {
                        const int num_halfspaces = std::min((int)module_data.static_obstacles[k].size(), _n_other_halfspaces);
                        for (int h = 0; h < num_halfspaces; h++) {
                            const int obs_id = (int)copied_obstacles.size() + h;
                            _a1[d][k][obs_id] = module_data.static_obstacles[k][h].A(0);
                            _a2[d][k][obs_id] = module_data.static_obstacles[k][h].A(1);
                            _b[d][k][obs_id] = module_data.static_obstacles[k][h].b;
                        }
                    }
                }
            }
        }
        void projectToSafety(const std::vector<DynamicObstacle> &copied_obstacles, int k, Vector2d &pos)
        {
            if (copied_obstacles.empty()) return;                                 /* there is no anchor */
            for (int iterate = 0; iterate < 3; iterate++)
                for (auto &obstacle : copied_obstacles) {
                    const double radius = _use_guidance ? 1e-3 : obstacle.radius;
                    dr_projection_.douglasRachfordProjection(pos, obstacle.prediction.modes[0][k - 1].position,
                                                             copied_obstacles[0].prediction.modes[0][k - 1].position,
                                                             radius + _cfg.robot_radius, pos);
                }
        }
        void setParameters(const RealTimeData &data, const ModuleData &, int k)
        {
            int constraint_counter = 0;
            if (k == 0) {
                for (int i = 0; i < _max_obstacles + _n_other_halfspaces; i++) {
                    setSolverParameterLinConstraintA1(0, _solver->_params, _dummy_a1, constraint_counter);
                    setSolverParameterLinConstraintA2(0, _solver->_params, _dummy_a2, constraint_counter);
                    setSolverParameterLinConstraintB(0, _solver->_params, _dummy_b, constraint_counter);
                    constraint_counter++;
                }
                return;
            }
            for (int d = 0; d < _n_discs; d++) {
                if (!_use_guidance) setSolverParameterEgoDiscOffset(k, _solver->_params, data.robot_area[d].offset, d);
                for (size_t i = 0; i < data.dynamic_obstacles.size() + _n_other_halfspaces; i++) {
                    setSolverParameterLinConstraintA1(k, _solver->_params, _a1[d][k][i], constraint_counter);
                    setSolverParameterLinConstraintA2(k, _solver->_params, _a2[d][k][i], constraint_counter);
                    setSolverParameterLinConstraintB(k, _solver->_params, _b[d][k][i], constraint_counter);
                    constraint_counter++;
                }
                for (int i = (int)data.dynamic_obstacles.size() + _n_other_halfspaces; i < _max_obstacles + _n_other_halfspaces; i++) {
                    setSolverParameterLinConstraintA1(k, _solver->_params, _dummy_a1, constraint_counter);
                    setSolverParameterLinConstraintA2(k, _solver->_params, _dummy_a2, constraint_counter);
                    setSolverParameterLinConstraintB(k, _solver->_params, _dummy_b, constraint_counter);
                    constraint_counter++;
                }
            }
        }
        std::shared_ptr<Solver> _solver;
    private:
        ModuleConfig _cfg;
        DouglasRachford dr_projection_;
        std::vector<std::vector<std::vector<double>>> _a1, _a2, _b;            /* [disc][k][row] */
        int _n_discs{1}, _max_obstacles, _n_other_halfspaces, _num_obstacles{0};
        bool _use_guidance{false};
        double _dummy_a1{1.}, _dummy_a2{0.}, _dummy_b{0.};
    };
#endif

#if SOLVER_NLIN > 0 && SOLVER_M > 0
    /* ---- guidance_constraints.h:32-50, 85-101 ---- */
    struct SolverResult
    {
        int exit_code; double objective; bool success; int guidance_ID; int color;
        void Reset() { success = false; objective = 1e10; exit_code = -1; guidance_ID = -1; color = -1; }
    };

    /* ---- guidance_constraints.cpp:264-434, batched ---- */
    class GuidanceConstraints
    {
    public:
        struct LocalPlanner
        {
            int id;
            std::unique_ptr<LinearizedConstraints> guidance_constraints;        /* keep the solver in the topology */
            std::unique_ptr<GUIDANCE_CONSTRAINTS_TYPE> safety_constraints;       /* avoid collisions (guidance_constraints.h:89) */
            std::shared_ptr<Solver> local_solver;                                /* distinct solver for each planner */
            SolverResult result;
            bool is_original_planner = false, disabled = true, taken = false, existing_guidance = false;
            LocalPlanner(int _id, const ModuleConfig &cfg, bool _is_original_planner = false) : id(_id), is_original_planner(_is_original_planner)
            {
                local_solver = std::make_shared<Solver>(_id + 1);                /* guidance_constraints.cpp:18-27 */
                guidance_constraints = std::make_unique<LinearizedConstraints>(local_solver, cfg);
                guidance_constraints->setTopologyConstraints();
                safety_constraints = std::make_unique<GUIDANCE_CONSTRAINTS_TYPE>(local_solver, cfg);
            }
        };

        GuidanceConstraints(std::shared_ptr<Solver> solver, const ModuleConfig &cfg) : _solver(solver), _cfg(cfg)
        {
            _use_tmpcpp = cfg.use_tmpcpp; _enable_constraints = cfg.enable_constraints;
            for (int i = 0; i < cfg.n_paths; i++) planners_.emplace_back(i, cfg);           /* :40-52 */
            if (_use_tmpcpp) planners_.emplace_back(cfg.n_paths, cfg, true);                 /* the non-guided planner */
        }
        /* stands in for global_guidance_->Update() + GetGuidanceTrajectory(i): the trajectories found for this tick */
        void setGuidanceTrajectories(const std::vector<GuidanceTrajectory> &t) { _guidance = t; mapGuidanceTrajectoriesToPlanners(); }   /* :106-108 */
        /* guidance_constraints.cpp:192-250 -- which planner continues which homotopy class.  Pass 1: a trajectory whose class equals
         * the guidance_ID of a planner's last result reserves that planner (existing_guidance: its previous MPC solution is a valid
         * warm start, consumed at :310).  Pass 2: the trajectories left over go to the planners left over -- with the reference's
         * quirk kept: its inner loop has no `break`, so the FIRST left-over trajectory claims every free planner (its entry ends at
         * the last one) and later left-over trajectories get none.  Integer bookkeeping: equal to modules.map_guidance_trajectories_to_planners. */
        void mapGuidanceTrajectoriesToPlanners()
        {
            for (auto &pl : planners_) { pl.taken = false; pl.existing_guidance = false; }
            _map_homotopy_class_to_planner.clear();
            std::vector<int> left_over;
            for (int i = 0; i < NumberOfGuidanceTrajectories(); i++) {
                size_t p = 0;
                while (p < planners_.size() && !(planners_[p].result.guidance_ID == _guidance[i].topology_class && !planners_[p].taken)) p++;
                if (p == planners_.size()) { left_over.push_back(i); continue; }
                _map_homotopy_class_to_planner[i] = (int)p;
                planners_[p].taken = planners_[p].existing_guidance = true;
            }
            for (int i : left_over)
                for (size_t p = 0; p < planners_.size(); p++)
                    if (!planners_[p].taken) { _map_homotopy_class_to_planner[i] = (int)p; planners_[p].taken = true; planners_[p].existing_guidance = false; }
        }
        const std::map<int, int> &guidanceToPlannerMap() const { return _map_homotopy_class_to_planner; }
        int NumberOfGuidanceTrajectories() const { return (int)_guidance.size(); }

        /* the OpenMP loop's body up to solve() (:281-337): false if the planner is disabled this tick */
        bool preparePlanner(LocalPlanner &planner, State &state, const RealTimeData &data, ModuleData &module_data)
        {
            planner.result.Reset();
            planner.disabled = false;
            if (planner.id >= NumberOfGuidanceTrajectories() && !planner.is_original_planner) { planner.disabled = true; return false; }
            auto &solver = planner.local_solver;
            *solver = *_solver;                                                            /* copy the main solver */
            if (planner.is_original_planner || !_enable_constraints) {
                planner.guidance_constraints->update(state, empty_data_, module_data);
                planner.safety_constraints->update(state, data, module_data);
            } else {
                if (_cfg.warmstart_with_mpc_solution && planner.existing_guidance) planner.local_solver->initializeWarmstart(state, _cfg.shift_previous_solution_forward);
                else initializeSolverWithGuidance(planner);
                planner.guidance_constraints->update(state, data, module_data);
                planner.safety_constraints->update(state, data, module_data);
            }
            for (int k = 0; k < _solver->N; k++) {
                if (planner.is_original_planner) planner.guidance_constraints->setParameters(empty_data_, module_data, k);
                else planner.guidance_constraints->setParameters(data, module_data, k);
                planner.safety_constraints->setParameters(data, module_data, k);
            }
            planner.local_solver->loadWarmstart();
            return true;
        }
        /* ANALYSIS AND PROCESSING (:343-360) */
        void recordResult(LocalPlanner &planner, int exit_code)
        {
            planner.result.exit_code = exit_code;
            planner.result.success = planner.result.exit_code == 1;
            planner.result.objective = planner.local_solver->_info.pobj;
            if (planner.is_original_planner) {
                planner.result.guidance_ID = 2 * _cfg.n_paths;
                planner.result.color = -1;
            } else {
                const GuidanceTrajectory &g = _guidance[planner.id];
                planner.result.guidance_ID = g.topology_class;
                planner.result.color = g.color;
                if (g.previously_selected) planner.result.objective *= _cfg.selection_weight_consistency;
            }
        }
        /* DECISION MAKING (:366-387) */
        int decide()
        {
            best_planner_index_ = FindBestPlanner();
            if (best_planner_index_ == -1) return planners_[0].result.exit_code;
            auto &best_planner = planners_[best_planner_index_];
            _solver->_output = best_planner.local_solver->_output;
            _solver->_info = best_planner.local_solver->_info;
            _solver->_params = best_planner.local_solver->_params;
            return best_planner.result.exit_code;
        }
        int optimize(State &state, const RealTimeData &data, ModuleData &module_data)
        {
            if (!_use_tmpcpp && _guidance.empty()) return 0;                                  /* :273-274 */
            std::vector<Solver *> active;
            std::vector<LocalPlanner *> active_planners;
            for (auto &planner : planners_)
                if (preparePlanner(planner, state, data, module_data)) { active.push_back(planner.local_solver.get()); active_planners.push_back(&planner); }
            const std::vector<int> codes = Solver::solveBatch(_batch, active);                 /* ONE launch instead of solver->solve() per thread (:339); every planner on its own slot */
            for (size_t i = 0; i < active.size(); i++) recordResult(*active_planners[i], codes[i]);
            return decide();
        }
        /* The UN-PATCHED drop-in: the reference's loop as it stands (guidance_constraints.cpp:279-361) -- every planner's own Solver solves on its
         * own handle and stream from its own OpenMP thread (`num_threads(8)`, no locks: SURVEY 8b threading contract).  Same results as optimize(),
         * bit for bit (a trajectory's result does not depend on what else runs); slower per tick, because every thread pays its own upload, launch
         * and download where optimize() pays one of each (tests/cpp/test_omp_solvers.cpp measures both). */
        int optimizeOpenMP(State &state, const RealTimeData &data, ModuleData &module_data, int num_threads = 8)
        {
            if (!_use_tmpcpp && _guidance.empty()) return 0;
            const int n = (int)planners_.size();
            (void)num_threads;
#pragma omp parallel for num_threads(num_threads)
            for (int i = 0; i < n; i++) {
                LocalPlanner &planner = planners_[i];
                if (!preparePlanner(planner, state, data, module_data)) continue;
                recordResult(planner, planner.local_solver->solve());                          /* :339 */
            }
            return decide();
        }
        void initializeSolverWithGuidance(LocalPlanner &planner)                               /* :390-414 */
        {
            auto &solver = planner.local_solver;
            const GuidanceTrajectory &g = _guidance[planner.id];
            for (int k = 1; k < solver->N; k++) {
                const Vector2d &cur_position = g.positions[k], &cur_velocity = g.velocities[k];
                solver->setEgoPrediction(k, "x", cur_position(0));
                solver->setEgoPrediction(k, "y", cur_position(1));
                solver->setEgoPrediction(k, "psi", std::atan2(cur_velocity(1), cur_velocity(0)));
                solver->setEgoPrediction(k, "v", std::sqrt(cur_velocity(0) * cur_velocity(0) + cur_velocity(1) * cur_velocity(1)));
            }
        }
        int FindBestPlanner()                                                                  /* :416-434 */
        {
            double best_solution = 1e10;
            int best_index = -1;
            for (size_t i = 0; i < planners_.size(); i++) {
                auto &planner = planners_[i];
                if (planner.disabled) continue;
                if (planner.result.success && planner.result.objective < best_solution) { best_solution = planner.result.objective; best_index = (int)i; }
            }
            return best_index;
        }
        std::vector<LocalPlanner> planners_;
        int best_planner_index_{-1};
        std::shared_ptr<Solver> _solver;
        BatchContext _batch;                                                     /* this module instance's batch: one state slot per local planner */
    private:
        std::map<int, int> _map_homotopy_class_to_planner;
        ModuleConfig _cfg;
        std::vector<GuidanceTrajectory> _guidance;
        RealTimeData empty_data_;
        bool _use_tmpcpp{true}, _enable_constraints{true};
    };
#endif

#if SOLVER_NSLK > 0
    /* ---- scenario_constraints.cpp:58-108, batched.  The Safe Horizon routine itself (ScenarioModule::optimize: scenario sampling,
     * polygon construction, support bookkeeping) lives in the external scenario_module; what it leaves in the solver are the
     * halfspace parameters of each parallel scenario solver, which is the input here. ---- */
    class ScenarioConstraints
    {
    public:
        struct ScenarioSolver
        {
            std::shared_ptr<Solver> solver;
            int exit_code{-1};
            std::vector<StaticObstacle> halfspaces;                                            /* [k] -> rows a.x <= b (+ slack) for this solver's scenario set */
            std::vector<std::vector<int>> scenario_of_row;                                     /* [k][row] -> scenario the row's sample belongs to (optional) */
            /* scenario_constraints.h:38-40: the scenarios of support of this solver's solution (those with an active row) */
            struct SupportSubsample { std::vector<int> scenarios; int active_rows{0}; int size() const { return (int)scenarios.size(); } } support;
            ScenarioSolver(int id) : solver(std::make_shared<Solver>(id)) {}
        };
        /* support of a solver's solution: rows with a.p_disc - (b + slack) >= -tol, distinct scenarios (host counterpart of tmpc_scenario_support) */
        void computeSupport(ScenarioSolver &s, double tol = 1e-6) const
        {
            s.support = {};
            for (int k = 1; k < _solver->N && k < (int)s.halfspaces.size() && k < (int)s.scenario_of_row.size(); k++) {
                const double psi = s.solver->getOutput(k, "psi"), slack = s.solver->getOutput(k, "slack");
                const double px = s.solver->getOutput(k, "x") + _disc_offset * std::cos(psi), py = s.solver->getOutput(k, "y") + _disc_offset * std::sin(psi);
                for (size_t j = 0; j < s.halfspaces[k].size() && j < s.scenario_of_row[k].size(); j++) {
                    const int scenario = s.scenario_of_row[k][j];
                    const auto &row = s.halfspaces[k][j];
                    if (scenario < 0 || row.A(0) * px + row.A(1) * py - (row.b + slack) < -tol) continue;
                    s.support.active_rows++;
                    if (std::find(s.support.scenarios.begin(), s.support.scenarios.end(), scenario) == s.support.scenarios.end()) s.support.scenarios.push_back(scenario);
                }
            }
        }
        ScenarioConstraints(std::shared_ptr<Solver> solver, int parallel_solvers, double disc_offset = 0.) : _solver(solver), _disc_offset(disc_offset)
        {
            for (int i = 0; i < parallel_solvers; i++) _scenario_solvers.emplace_back(new ScenarioSolver(i));   /* :18-26 */
        }
        /* scenario_module.setParameters(data, k) (:76-79): ego disc offset, k = 0 and unused slots the dummies (1, 0, x + 100) */
        void setParameters(ScenarioSolver &s, double state_x, int k)
        {
            auto &p = s.solver->_params;
            s.solver->setParameter(k, "ego_disc_0_offset", _disc_offset);
            const StaticObstacle *rows = (k >= 1 && k < (int)s.halfspaces.size()) ? &s.halfspaces[k] : nullptr;
            for (int j = 0; j < SOLVER_NSLK; j++) {
                const bool live = rows && j < (int)rows->size();
                s.solver->setParameter(k, "disc_0_scenario_constraint_" + std::to_string(j) + "_a1", live ? (*rows)[j].A(0) : 1.0);
                s.solver->setParameter(k, "disc_0_scenario_constraint_" + std::to_string(j) + "_a2", live ? (*rows)[j].A(1) : 0.0);
                s.solver->setParameter(k, "disc_0_scenario_constraint_" + std::to_string(j) + "_b", live ? (*rows)[j].b : state_x + 100.0);
            }
            (void)p;
        }
        int optimize(State &state, const RealTimeData &, ModuleData &)
        {
            std::vector<Solver *> batch;
            for (auto &solver : _scenario_solvers) {
                *solver->solver = *_solver;                                                    /* copy the main solver */
                for (int k = 0; k < _solver->N; k++) setParameters(*solver, state.get("x"), k);
                solver->solver->loadWarmstart();                                               /* load the previous solution */
                batch.push_back(solver->solver.get());
            }
            const std::vector<int> codes = Solver::solveBatch(_batch, batch);                  /* scenario_module.optimize(data) of every solver: one launch */
            for (size_t i = 0; i < codes.size(); i++) { _scenario_solvers[i]->exit_code = codes[i]; computeSupport(*_scenario_solvers[i], _support_tolerance); }
            double lowest_cost = 1e9;                                                           /* :93-107 */
            _best_solver = nullptr;
            for (auto &solver : _scenario_solvers)
                if (solver->exit_code == 1 && solver->solver->_info.pobj < lowest_cost) { lowest_cost = solver->solver->_info.pobj; _best_solver = solver.get(); }
            if (_best_solver == nullptr) return _scenario_solvers.front()->exit_code;
            _solver->_output = _best_solver->solver->_output;
            _solver->_info = _best_solver->solver->_info;
            _solver->_params = _best_solver->solver->_params;
            return _best_solver->exit_code;
        }
        std::vector<std::unique_ptr<ScenarioSolver>> _scenario_solvers;
        ScenarioSolver *_best_solver{nullptr};
        std::shared_ptr<Solver> _solver;
        double _support_tolerance{1e-6};
        BatchContext _batch;                                                     /* this module instance's batch: one state slot per scenario solver */
    private:
        double _disc_offset;
    };
#endif
}
#endif
