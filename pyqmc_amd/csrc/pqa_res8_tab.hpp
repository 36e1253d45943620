// pqa_res8_tab.hpp — launch geometry and the table struct of the second-generation resident sweep (pqa_res8.hpp): what the handle
// (pqa_internal.hpp) and the host side (pqa_res8.hip) need.  The kernel itself is compiled by pqa_res8.hip only.
#pragma once
#include "pqa_res.hpp"

#define PQA_R8_NT 256
#define PQA_R8_NW 8
#define PQA_R8_MAXQ 32   // k-steps one wave contracts at most (kt <= 4 * 32 * KW)
#define PQA_R8_WS 32      // doubles per walker of the per-walker scalars: 0..15 as k_sweep_res; 16..19 U, grad U of the decided electron at its
                          // proposal; 20..23 the same of the NEXT electron at its current position if the move is rejected, 24..27 if it is accepted

struct R8Tab {
  int kt;                   // tile rows: the AOs in their own order, padded to x4 (coefficient copy d_cres[s] [kt][ldc])
  int cstride;              // doubles between the five component planes of the tile (8 kt, padded so that planes c and c + 1 start 128 B apart mod 256)
  int nitem;                // AO work items
  int wave_off[5];          // items of wave w: [wave_off[w], wave_off[w + 1])
  const int* item_hdr;      // [nitem][4]: l, primitives, first primitive in the deduplicated tables, 0
  const int* item_lane;     // [nitem][8][2]: atom of the slot (-1: idle), tile row of the shell's first function
  const double* item_xyz;   // [nitem][8][3]: the slot's atom position (one LDS round trip per item instead of atom index -> position)
  int nprim_u;
  const double* prim_exp_u;
  const double* prim_coef_u;
  int region;               // doubles of the tile / partial-sum / orbital-row region
  int jstage;               // offset (doubles) of the Jastrow sums' staging area [8][12][33] in the region, behind the partials and orbital rows
  int stagger;              // the block that shares its CU with an earlier one (LDS base > 0) starts this many x 3 us late: the two blocks' phases
                            // (AO / contraction / sums) then interleave instead of running in lock step (PQA_R8_STAGGER)
  double* xaos;             // (per launch) the walker-major coordinates [W][N][3] the ECP passes read, written with the planes at the end of the sweep
                            // when an energy evaluation follows (instead of a transpose launch), or nullptr
  int abl;                  // timing builds (-DPQA_RES_CLK) only: phases left out, PQA_R8_ABL bit mask (1 AO, 2 contraction, 4 Jastrow, 8 row / tape prefetch, 16 cache-row stores)
};
#ifdef PQA_RES_CLK
#define PQA_R8_ON(bit) (!(RT.abl & (bit)))
#else
#define PQA_R8_ON(bit) true
#endif
__host__ __device__ inline size_t r8_lds_fixed(int nprim, int natom, int na, int nitem) {
  const size_t d = PQA_R8_NW * 32 + PQA_R8_NW * PQA_R8_WS + 24 * (size_t)nitem + 2 * (size_t)nprim + 3 * (size_t)natom + 2 * (size_t)natom * (na > 0 ? na : 1) +
                   2 * (size_t)natom * PQA_JQP + PQA_RES_JT;
  const size_t i = 20 * (size_t)nitem + 64 + 8;
  return d * sizeof(double) + i * sizeof(int);
}

