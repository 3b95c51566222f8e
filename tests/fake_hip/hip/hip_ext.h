// hip_ext.h -- TEST INFRASTRUCTURE (see hip_runtime.h in this directory)
#pragma once
#include "hip_runtime.h"
