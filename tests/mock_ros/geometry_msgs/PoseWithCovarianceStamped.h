// Minimal stand-in for <geometry_msgs/PoseWithCovarianceStamped.h> (TEST ONLY): see Pose.h.
#pragma once
#include "Pose.h"
