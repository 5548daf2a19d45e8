/* mpc_planner_types/module_data.h -- restated after the reference's module_data.h:21-34.  The shared path object is a
 * RosTools::Spline2D there (absent); here the contouring segments themselves. */
#ifndef MODULE_DATA_HIP_H
#define MODULE_DATA_HIP_H

#include <vector>

#include <mpc_planner_types/data_types.h>

namespace MPCPlanner
{
    struct ModuleData
    {
        std::vector<StaticObstacle> static_obstacles;               /* [k] -> halfspaces */
        std::vector<PathSegment> path;                              /* segments from the closest one on (contouring.cpp:94-124) */
        int current_path_segment{-1};
        void reset() { static_obstacles.clear(); path.clear(); current_path_segment = -1; }
    };
}
#endif
