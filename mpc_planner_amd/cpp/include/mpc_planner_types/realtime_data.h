/* mpc_planner_types/realtime_data.h -- restated after the reference's realtime_data.h:16-51 (the members the path reads). */
#ifndef MPC_REALTIME_DATA_HIP_H
#define MPC_REALTIME_DATA_HIP_H

#include <chrono>
#include <vector>

#include <mpc_planner_types/data_types.h>

namespace MPCPlanner
{
    struct RealTimeData
    {
        std::vector<Disc> robot_area;
        std::vector<DynamicObstacle> dynamic_obstacles;
        Vector2d goal;
        bool goal_received{false};
        double intrusion{0.};
        std::chrono::system_clock::time_point planning_start_time;

        RealTimeData() = default;
        void reset()                                                /* :37-47: the robot area survives a reset */
        {
            std::vector<Disc> robot_area_copy = robot_area;
            *this = RealTimeData();
            robot_area = robot_area_copy;
            goal_received = false;
        }
    };
}
#endif
