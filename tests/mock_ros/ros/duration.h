// Minimal stand-in for <ros/duration.h> (TEST ONLY).
#pragma once
#include "time.h"
