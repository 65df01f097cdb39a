// Minimal stand-in for <geometry_msgs/TransformStamped.h> (TEST ONLY): see Pose.h.
#pragma once
#include "Pose.h"
