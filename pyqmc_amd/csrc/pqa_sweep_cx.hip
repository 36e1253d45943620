// pyqmc_amd C ABI implementation (host side): step / flush launches of the lane-per-walker sweep for COMPLEX determinants.
#include "pqa_sweep_launch.hpp"

void launch_step_cx(pqa_handle* h, const LwState& L, const MoveBuf& mb, const StepArgs& a, int rowlen) {
  if (h->S.pbc) launch_step_lw<true, true>(h, L, mb, a, rowlen); else launch_step_lw<false, true>(h, L, mb, a, rowlen);
}
void launch_flush_cx(pqa_handle* h, const LwState& L, int s, long W, long w0, long w1, int j_lo, int j_hi, int nq, int rowlen, int n_s) {
  launch_flush_lw<true>(h, L, s, W, w0, w1, j_lo, j_hi, nq, rowlen, n_s);
}
