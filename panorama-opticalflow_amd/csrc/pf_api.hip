// C-ABI implementation (include/panoflow.h): context, HBM arena, orchestration of the kernel families
// on HIP streams.  Host code only launches kernels and moves data; all pixel arithmetic is in the
// kernels_*.hip files.  There is no CPU fallback anywhere in this library.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/panoflow.h"
#include "pf_common.hpp"

using namespace pf;


// One translation unit in six parts (the anonymous namespace and the extern "C" block span several of them):
#include "pf_api_ctx.inl"      // pf_ctx, errors / warnings, arena, profiling events, geometry, checks
#include "pf_api_solve.inl"    // run_level, solve buffers, solve / solve_n (stream orchestration)
#include "pf_api_life.inl"     // finish / CallGuard, pf_create* / pf_destroy, memory helpers            (opens extern "C")
#include "pf_api_entry.inl"    // pf_flow* / pf_blend* / pf_novel_view* incl. the throughput mode
#include "pf_api_stitch.inl"   // pf_stitch_*
#include "pf_api_stage.inl"    // pf_stage_*, pf_profile_*, pf_level_pixels / pf_algorithmic_bytes
}  // extern "C"

