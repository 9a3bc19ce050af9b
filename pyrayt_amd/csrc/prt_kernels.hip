// prt_kernels.hip -- HIP kernels (gfx950 / CDNA4) + the C-ABI of include/prt.h.
//
// One translation unit, in this order:
//   prt_device.hpp   per-ray arithmetic: primitives, CSG nodes, normals, shading, the step interpreter
//   prt_scene.hpp    host side: scene object, scene compiler, per-device upload
//   prt_trace_kernels.hpp   trace kernels: k_generation (fused PROPAGATE + INTERACT + record, decoupled
//                    look-back compaction), the three-kernel path k_hit / k_scan / k_shade /
//                    k_advance, the per-object kernels k_intersect / k_normals / k_material_trace
//   prt_sources.hpp  k_source (ray sources on the device)
//   prt_render.hpp   k_render, k_render_hits, k_gooch*, k_camera, k_edge_* (renderers)
//   prt_ops.hpp      kernels of tinygfx.g3d.operations / primitives as functions
//   prt_abi_states.hpp      extern "C": per-state entry points (+ prt_host_shade.hpp: a caller's own Material.trace())
//   prt_trace_runtime.hpp   extern "C": prt_trace*, the ticket runtime
//   prt_abi_render_ops.hpp  extern "C": renderers, operations
//   prt_gather.hpp, prt_frame.hpp   extern "C" + kernels: frame re-assembly across ranks, reductions over the frame
//
// Data layout in HBM
//   ray set      (13, n) float64 row-major = PyRayT's RaySet verbatim (pyrayt/_pyrayt.py:13-44):
//                one contiguous row per field -> lane i of a wave reads element i of every
//                row: each of the 13 loads is a fully coalesced 512 B wave transaction.
//   record rows  (15, cap) float64 row-major: one contiguous row per DataFrame column
//                (_pyrayt.py:154-165); generation g appends its rows at column row_base(g).
//   scene        DevPrim[] + DevInstr[]: wave-uniform, fetched through the scalar cache.
//   per-lane CSG hit lists live in LDS (see prt_device.hpp), never in HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/prt.h"
#include "prt_device.hpp"

#include "prt_scene.hpp"

#include "prt_trace_kernels.hpp"

#include "prt_sources.hpp"
#include "prt_render.hpp"
#include "prt_ops.hpp"

// deep CSG trees may need more than the default 64 KiB of dynamic LDS per workgroup
static int raise_lds_limits() {
  HIP_TRY(hipFuncSetAttribute((const void*)k_render, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_render_hits, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_intersect, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  return PRT_OK;
}


#include "prt_abi_states.hpp"
#include "prt_trace_runtime.hpp"
#include "prt_abi_render_ops.hpp"

// ------------------------------------------------------------------------------------------------
// frame re-assembly across ranks (RCCL all-gather + placement kernel)
// ------------------------------------------------------------------------------------------------
#include "prt_gather.hpp"

// ------------------------------------------------------------------------------------------------
// reductions over the result frame
// ------------------------------------------------------------------------------------------------
#include "prt_frame.hpp"
