// k_planner.hip - the baseline planners: GPMP2 / RRT-Connect kernels (planner.hpp) and their C ABI (planner_host.hpp).
#include "host.hpp"

using namespace mpdx;

#include "planner_host.hpp"
