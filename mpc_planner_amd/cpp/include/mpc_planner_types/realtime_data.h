/*
 * mpc_planner_types/realtime_data.h (HIP flavour) -- what the accelerated path reads of the per-tick sensor data; written
 * after mpc_planner_types/include/mpc_planner_types/realtime_data.h:16-51 (member names and reset() semantics kept so that module
 * code compiles unchanged; costmap, reference path, road bounds and the past trajectory are not on this path and stay with the
 * reference's own header in a full tree).
 */
#ifndef MPC_REALTIME_DATA_HIP_H
#define MPC_REALTIME_DATA_HIP_H

#include <chrono>
#include <utility>
#include <vector>

#include <mpc_planner_types/data_types.h>

namespace MPCPlanner
{
    struct RealTimeData
    {
        // consumers on the path:
        std::vector<DynamicObstacle> dynamic_obstacles;                       // Ellipsoid-/LinearizedConstraints::setParameters, update
        std::vector<Disc> robot_area;                                         // ego_disc_<d>_offset parameters
        std::chrono::system_clock::time_point planning_start_time;           // solver_timeout bookkeeping of the optimize() loops
        Vector2d goal;                                                        // GoalModule (generated solvers)
        bool goal_received{false};
        double intrusion{0.};                                                 // feedback value published by the ROS wrappers

        // Everything but the robot's disc model is per-tick data (reference :37-47).
        void reset()
        {
            RealTimeData fresh;
            fresh.robot_area = std::move(robot_area);
            *this = std::move(fresh);
        }
    };
}
#endif
