// Shared constants of the matching kernels (match.hip, match_top2.hip).
#pragma once
#include "common.h"

#define KP REFVSR_MATCH_KP            // halfs per row (152)
#define ROWB (KP * 2)                 // bytes per row (304)
#define COLB REFVSR_MATCH_COLBLOCK    // LR columns per workgroup
#define KSTEPS 9
