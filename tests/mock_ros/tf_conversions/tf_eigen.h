// Minimal stand-in for <tf_conversions/tf_eigen.h> (TEST ONLY): included by the odometry nodelet, nothing of it used.
#pragma once
