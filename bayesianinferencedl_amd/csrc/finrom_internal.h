// Internal declarations of the reduced model and the learned error model (on top of finrom_core.h).
#pragma once
#include "finrom_core.h"

namespace finrom {

// ---- learned error model (mlp_kernels.hip, finrom_mlp_*) ----------------------------------
struct MlpDev {
  int n_in, n_w, n_layers, n_out;
  const float* W0; const float* b0;            // [n_in x n_w], [n_w]
  const float* scale; const float* shift;      // [(n_layers + 1) x n_w] inference-form batch normalisation (last row: the head)
  const float* W; const float* b;              // [n_layers x n_w x n_w], [n_layers x n_w]
  const float* Wh; const float* bh;            // [n_w x n_out], [n_out]
};
// the error model's forward pass riding as ONE MORE workgroup per sample in rom_small_proj_kernel (finrom_romml_grad's one-sample
// form: the ROM's contraction does not depend on it, the solve kernel behind both is the first to need its output)
struct MlpFuse {
  int on = 0;
  MlpDev m{};
  const double* k = nullptr; const double* data = nullptr; int64_t data_stride = 0;
  float* tape = nullptr; double* e_out = nullptr; double* data_shift = nullptr;
  // the sub-fin averages theta = S k too: every contraction workgroup forms its sample's theta itself (the same sums in the
  // same order, into its own 16 doubles of theta_scr [S x NC x 16], which its scalar loads then read), workgroup 0 also writes
  // theta_out [S x P] for the kernels behind
  const double* Sop = nullptr; int P = 0; double* theta_scr = nullptr; double* theta_out = nullptr;
  // finrom_hmc_leapfrog: the position update rides in front -- every reader of the field sees k + eps * mom, and the network's
  // workgroup (which reads all of it anyway) writes the moved field to k_out (NOT in place: the sample's other workgroups read k)
  const double* mom = nullptr; double eps = 0.0; double* k_out = nullptr;
  // round 4: the first layer over nw0 spare workgroups (partial sums y0_part [S x nw0 x 64] floats), the layers behind it as a
  // spare wave of the solve kernel (rom_onesample.hip)
  int nw0 = 4; float* y0_part = nullptr;
  // finrom_hmc_leapfrog, steps behind the first of a trajectory: theta = S (k + eps mom) arrives as HMC_THETA_PARTS partial sums
  // [S x HMC_THETA_PARTS x 16] left by the previous step's momentum update (which forms the new momentum anyway) -- the 4.6 us theta
  // phase at the head of every contraction workgroup becomes one load of 8 x P doubles
  const double* theta_parts = nullptr;
};
constexpr int HMC_THETA_PARTS = 8;      // = mlp_kernels.hip's MLP_SPLIT: one partial sum per workgroup of the backward kernel
constexpr int HMC_THETA_MAXP = 10;      // averages the carry handles (the fin has nine); beyond: every step forms theta itself
// finrom_hmc_leapfrog: the momentum update rides behind the gradient (mlp_backward_kernel's epilogue):
//   dU = (kq - mean) + coef * grad  (0 for a flagged sample);  mom -= eps_cpri * dU
struct HmcTail {
  int on = 0;
  const double* kq = nullptr; const double* mean = nullptr; double* mom = nullptr; double* dU = nullptr;
  double coef = 0.0, eps_cpri = 0.0; const int* info = nullptr;
  // the NEXT step's sub-fin averages while the new momentum is at hand: theta_parts[s][wg][p] = sum over this workgroup's rows of
  // Sop[p][i] (kq[i] + eps mom_new[i]); nullptr: not wanted
  double eps = 0.0; double* theta_parts = nullptr;
};
int launch_mlp_forward(const MlpDev& m, const double* k, int64_t S, const double* data, int64_t data_stride, float* tape,
                       double* e_out, double* data_shift, hipStream_t st, const double* Sop = nullptr, int P = 0,
                       double* theta_out = nullptr);
// (g_parts, n_parts: instead of g_theta, its n_parts partial sums [S x n_parts x 32] as rom_grad_contract_small_kernel leaves them)
int launch_mlp_backward(const MlpDev& m, int64_t S, const float* tape, const double* data, int64_t data_stride, const double* qoi_r,
                        const double* e_nn, const double* g_theta, const double* Sop, int P, double* grad, hipStream_t st,
                        const double* g_parts = nullptr, int n_parts = 0, const float* g0_in = nullptr, const HmcTail* tail = nullptr);
// (g0_in [S x 64]: the walk back through the head and the hidden layers has been done -- MlpBackFuse -- and left its result there)
struct MlpBackFuse {             // that walk as one more workgroup per sample of rom_grad_contract_small_kernel
  int on = 0;
  MlpDev m{};
  const float* tape = nullptr; const double* data = nullptr; int64_t data_stride = 0;
  const double* qoi_r = nullptr; const double* e_nn = nullptr; float* g0_out = nullptr;
};

// ---- HMC trajectory bookkeeping (hmc_kernels.hip, finrom_hmc_*) ------------------------------------
struct HmcDev {                  // finrom_hmc_state with the host-side fields resolved
  int64_t C; int n;
  double eps, c_lik, c_pri;
  const double* mean; double* K; double* U; double* dU;
  double* Kq0; double* P; double* dUq; double* H0;
  const double* P_block; const double* lu_block;
  long long* jt; long long* pt; long long* accept;
  double* trace; const double* loss; const int* info;
};
int launch_hmc_begin(const HmcDev& h, hipStream_t st);
int launch_hmc_end(const HmcDev& h, const double* Kq, hipStream_t st);

// ---- ROM ------------------------------------------------------------------------------
constexpr int ROM_MAX_PHASES = 8;
struct RomDev {
  int n, r, rp, NB, P, n_obs;           // rp = 16*NB padded basis size
  int solve_in_lds;                     // packed factor fits in LDS (rp <= 176)
  long long* trace;                     // FINROM_TRACE (see FomDev::trace)
  int clock_probe;                      // FINROM_CLOCK_PROBE: a few workgroups print their shader clock (diagnostic)
  // psi tables, rows grouped 4 per k-step and sorted by term count into phases of constant NT
  int n_phases;
  int phase_nt[ROM_MAX_PHASES], phase_ks0[ROM_MAX_PHASES], phase_ks1[ROM_MAX_PHASES], phase_slot0[ROM_MAX_PHASES];
  // the same tables cut into LDS-sized chunks of whole k-steps (LDS-staged kernel, NB <= 5)
  int n_chunks;
  const int* ch_nt; const int* ch_nks; const int* ch_off; const int* ch_bytes;   // [n_chunks]; ch_off in doubles
  const double* tvc;                    // chunk images: values then theta indices, each padded to 1 KiB
  // the same rows regrouped for the single-wave kernels so that the four rows of a k-step share ONE list of theta indices
  // (rows with the same term pattern together; leftovers merged under the union of their patterns, absent terms = zero
  // rows): theta then comes from SGPRs, and the k-step needs no theta-index loads, no LDS reads and no per-lane addresses
  int n_uphases;
  int uphase_nt[ROM_MAX_PHASES], uphase_ks0[ROM_MAX_PHASES], uphase_ks1[ROM_MAX_PHASES], uphase_slot0[ROM_MAX_PHASES];
  const double* tvu;                    // [(nuslots + 4) * 4 * rp]
  const int* kpat;                      // [nuslots + 16] theta index of each slot (0 = the constant 1)
  int tvu_bytes;
  int nku;                              // number of pattern-uniform k-steps (even; sorted by term count, descending; every
                                        // term-count group holds an even number of k-steps, padded with a zero k-step)
  int uend[4];                          // uend[t] = number of k-steps with more than t terms (uend[0] = nku)
  const int* kmeta;                     // [(nku + 8) * 8] per k-step: first slot, term count, 2 x padding, theta indices of its 4 slots
  // the pattern-uniform k-steps grouped by their leading parameter and accumulated divided by it (proj_main_grouped; built in
  // finrom_rom_create, where the form is described): records {first slot, terms, flags (1: the first coefficient is 1, 2: rescale
  // the accumulators first), ext index of that factor, ext indices of the coefficients}
  const double* tvg; int tvg_bytes;
  const int* kmg;                       // [(nkg + 8) * 8]
  int nkg;                              // multiple of 3; 0: no grouped form (nothing to group, too many parameters, NB > 5)
  int n_ext, ext_final;                 // per-sample scalars (<= 64); ext index of the factor behind the last k-step
  const int* ext_def;                   // [n_ext * 3] ext[l] = (theta'[a] / theta'[b]) ^ (1 + sq), theta'[0] = 1
  double* ext;                          // [S x n_ext] workspace of the CALL (set by rom_project; nullptr: the ungrouped loop)
  const double* tv;                     // [(nslots + 4) * 4 * rp]  padded r-vectors, slot-major
  const int* pidx;                      // [(nslots + 4) * 4]       theta index of each r-vector (0 = constant 1)
  // rows with a non-zero load F (root nodes), same slot format with a runtime term count
  int rhs_nk, rhs_nt;
  const double* rhs_tv; const int* rhs_pidx; const double* rhs_f;
  const double* obs_phi;                // [n_obs x r]
};
struct RomGradArgs {                      // adjoint-gradient stage of rom_solve_kernel (finrom_rom_grad)
  const double* data = nullptr; int64_t data_stride = 0;    // observations [n_obs] (stride 0) or [S x n_obs]
  const double* theta = nullptr;                            // [S x P]
  double* J = nullptr; double* g = nullptr;                 // [S], [S x P]
  int npairs = 0; const int* pair_p = nullptr; const int* pair_i = nullptr; const double* Gt = nullptr;
  double* gpart = nullptr; int* ticket = nullptr;   // small batches (rom_grad_contract_small_kernel): [S x NG x 32] partial sums, [S] arrival counters (zero between calls)
  int defer_sum = 0;        // (small batches) leave the NG partial sums in gpart: the consumer adds them (mlp_backward_kernel in finrom_romml_grad)
  int info_store = 0;       // (one-sample solve kernel) info[s] is stored (0 or 2), not or-ed into: the caller did not have to clear it
  double* vw = nullptr;     // scratch [S x 2 rp]: when set, the substitution kernel leaves v_r and w_r there and the
                            // contraction g_i = sum_p theta_p v_r^T G_pi w_r runs in rom_grad_contract_kernel (fp64 MFMA, 16 samples per wave)
};
// offline/online form of the reduced operator (finrom_rom_set_gram, rom_gram.hip)
constexpr int ROM_GRAM_MAX_PAIRS = 64;
struct RomGramDev {
  int npairs = 0;
  const int* pair_p = nullptr; const int* pair_q = nullptr;   // [npairs], 0 = the constant term
  const double* Gt = nullptr;     // [npairs][NB(NB+1)/2][64][4]  tiles in the MFMA C/D register layout (r <= 96)
  const double* Gp = nullptr;     // [npairs][rp(rp+1)/2]         packed upper triangle, row by row (r > 96)
  const double* h = nullptr;      // [P+1][rp]                    h_p = Psi_p^T F
};
int launch_rom_gram(const RomDev& p, const RomGramDev& gm, const double* theta, int64_t S, double* Ar, double* Br, int factor,
                    int* info, hipStream_t st, double* w_r = nullptr, double* qoi_r = nullptr);
int launch_rom_grad(const RomDev& p, const double* Ar, const double* Br, int64_t S, double* w_r, double* qoi_r,
                    int* info, const RomGradArgs& ga, hipStream_t st);
int launch_rom_grad_contract(const RomDev& p, int64_t S, const RomGradArgs& ga, hipStream_t st);
int launch_rom_chol_blocked(const RomDev& p, double* Ar, int64_t S, int* info, hipStream_t st);
int launch_rom_proj(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info, hipStream_t st,
                    double* w_r = nullptr, double* qoi_r = nullptr, int* cu_ticket = nullptr);
constexpr int ROM_SPLITK_MAX_S = 64;      // batches up to this size take the split-K projection kernel (r = 49..96)
// (r mod 16 in 1..8, NB >= 7: the HalfCover form of the multi-wave kernels, rom_proj_half.hip)
int launch_rom_proj_half(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                         hipStream_t st, double* w_r, double* qoi_r);
int launch_rom_proj_wide(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                         hipStream_t st, double* w_r, double* qoi_r);
bool rom_splitk_applies(const RomDev& p, int64_t S);
// one-sample call patterns (rom_onesample.hip): contraction over several workgroups per sample + MFMA-form solve kernel
bool rom_onesample_applies(const RomDev& p, int64_t S);
size_t rom_onesample_scratch_bytes(const RomDev& p, int64_t S);
int launch_rom_onesample(const RomDev& p, const double* theta, int64_t S, double* part, int grad, const RomGradArgs& ga, double* w_r,
                         double* qoi_r, int* info, hipStream_t st, const MlpFuse* fuse = nullptr);
constexpr int ROM_GRAD_SMALL_NG = 36;      // blocks per sample in rom_grad_contract_small_kernel
int launch_rom_grad_contract_small(const RomDev& p, int64_t S, const RomGradArgs& ga, hipStream_t st, const MlpBackFuse* bf = nullptr);
int launch_rom_proj_splitk(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                           hipStream_t st, double* w_r, double* qoi_r, const RomGradArgs& ga);
int launch_rom_proj_single(const RomDev& p, const double* theta, int64_t S, double* Ar, double* Br, int factor, int* info,
                           hipStream_t st, double* w_r, double* qoi_r, int* cu_ticket);   // cu_ticket: 4096 ints of device scratch or nullptr
int launch_rom_solve(const RomDev& p, const double* Ar, const double* Br, int64_t S, double* w_r,
                     double* qoi_r, double* Ar_out, double* Br_out, int* info, int factored, hipStream_t st);

}  // namespace finrom
