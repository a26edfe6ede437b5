// fan_diag.h -- the two diagnostic hooks of the fan kernels; both compile to nothing in the product.
//   -DFAN_STAMPS      wave w of a tile stores the low word of s_memtime after each phase into the tile's slice of the optional alpha
//                     output ([R,20] floats: slot 16 w + k), which then carries no alphas (scripts/fan_stamps.py reads the timeline)
//   -DFAN_EXIT_AFTER=k  every tile stops after the phase that ends at stamp k: the vector-instruction count of a phase is the
//                     difference of two such builds' SQ_INSTS_VALU (scripts/pmc_fan_phases.sh)
#pragma once
#ifdef FAN_STAMPS
#define STAMP(k) do { if (a.alpha && lane == 0 && n_live == FR) reinterpret_cast<uint32_t*>(a.alpha)[ray0 * FS + 16 * wave + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
#ifdef FAN_EXIT_AFTER
#define FAN_EXIT(k) do { if (FAN_EXIT_AFTER == (k)) return; } while (0)
#else
#define FAN_EXIT(k) do { } while (0)
#endif
#if defined(FAN_STAMPS) && FAN_STAMPS == 2
#define ESTAMP(k) STAMP(k)
#else
#define ESTAMP(k) do { } while (0)
#endif

