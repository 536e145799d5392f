// sf_layout.h -- MFMA operand image of a flow ("packed" layout) and the kernel descriptor.
//
// Tile convention (v_mfma_f32_32x32x2_f32, one wave = 64 lanes):
//   lane l:  c = l & 31  (sample column of a 32-sample tile),  h = l >> 5 (row half)
//   an activation tile is 16 VGPRs a[0..15]; a[r] on lane (c,h) holds feature
//       row(r,h) = (r & 3) + 8*(r >> 2) + 4*h          of sample c
//   which is exactly the C/D register map of the 32x32 MFMA, so the accumulator of one
//   layer is consumed register-by-register as the B operand (K pair = {row(r,0), row(r,1)})
//   of the next layer: activations never leave registers and never change lanes.
//   A "group" g = registers 4g..4g+3 = rows 8g..8g+7; weights are stored one float4 per
//   lane per (output tile, input group):
//       W4[(mt*nG + kg)*64 + l] = { W[o][i_j] : j=0..3 },  o = orow[mt*32 + (l&31)],
//                                  i_j = irow[kg*8 + 4*(l>>5) + j]
//   so one coalesced 1 KiB load feeds four MFMAs (x NS sample tiles).
//   Bias image: [mt][h][16] = b[orow[mt*32 + row(r,h)]]  (accumulator initial value).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/synference_hip.h"

#define SF_DMAX 16
#define SF_NBMAX 4

struct SfDev {
  const float* packed;   // forward operand image, all transforms
  const float* packedT;  // transposed operand image (backward data-gradient), training only
  const float* cst;      // constants image (see c_* offsets)
  int kind, D, C, H, T, K, NB, scale_fn;
  int HT;    // hidden tiles = ceil(H/32)
  int PT;    // NSF: tiles of spline parameters per dim pair
  int KMAX;  // NSF: bins capacity of the PT layout
  int JP;    // NSF: dim pairs per transform (max over parity)
  int nGu, nGc, nGh;  // active input groups: u tile, context, hidden
  int t_stride;       // floats per transform in packed / packedT
  // LDS staging plan of one transform's image: part p = floats [part_off[p], part_off[p+1]); parts are
  // whole layer blocks in execution order; n_parts == 0 -> some block exceeds the LDS budget (stream from L2)
  int n_parts, part_off[5], part_max;
  int blk_part[SF_NBMAX], head_part;  // NSF: part holding block k / the spline head
  // bf16 operand image (hidden_bf16 != 0): part p also stages elements [partB_off[p], partB_off[p+1]) of it, right behind its
  // fp32 slice in LDS; part_bytes_max = the largest part, both slices (a single-part image: {0, tB_stride})
  int partB_off[5], part_bytes_max;
  // offsets in floats relative to the transform base -------------------------------------
  int o_w0, o_wc, o_b0;                  // MAF initial (u part, context part, b0+bc)
  int o_wk[SF_NBMAX], o_bk[SF_NBMAX];    // MAF hidden blocks
  int o_wf, o_bf;                        // MAF final
  int o_hv, o_hvb;                       // MAF head rows for the VALU form: [slot][a|m][half][HT*16], bias [slot][a|m]
  int o_winu, o_winc, o_bin;             // NSF initial
  int o_wg[SF_NBMAX], o_bg[SF_NBMAX];    // NSF GLU gate
  int o_w1[SF_NBMAX], o_b1[SF_NBMAX], o_w2[SF_NBMAX], o_b2[SF_NBMAX];
  int o_wout, o_bout;                    // NSF spline head, [JP][PT] tiles
  int o_lu;                              // NSF LU block: L[D*D] U[D*D] udiag[D] bias[D]
  // transposed image (data-gradient operands W^T), offsets relative to t*tT_stride ---------
  int tT_stride;
  int nGf;                               // MAF: active groups of the final-layer output tile
  int oT_wf, oT_wk[SF_NBMAX], oT_w0;     // MAF
  int oT_wout, oT_w1[SF_NBMAX], oT_w2[SF_NBMAX], oT_winu;  // NSF (oT_wout: [JP] blocks)
  int oT_wc;                // context-gradient operands: MAF Wc^T / NSF Win_c^T, [ceil(C/32)] tiles
  int oT_wg[SF_NBMAX];      // NSF gate Wg^T
  // MAF degree-sorted hidden layout (sf_layout.cpp): units of MADE degree g (1..D-1) occupy whole
  // rows of ONE tile g_tile[g]; covering every unit of degree <= g takes g_kend[g] input groups.
  // inc_ok = 1 when no degree group straddles a tile (enables the incremental inverse).
  int inc_ok;
  int g_tile[SF_DMAX], g_kend[SF_DMAX];  // g_tile = LAST tile of the group
  int g_lo[SF_DMAX];                      // FIRST tile of the group (== g_tile for aligned layouts)
  int mt_kend[4];  // input groups needed by hidden output tile mt (== nGh when not degree-sorted)
  // bf16 operand image of the hidden HxH layers (inference, opt-in): [mt][ks][lane][8] bf16, ks = 16-row steps
  const unsigned short* packedB;
  int hidden_bf16, nKS, tB_stride;            // nKS = ceil(nGh/2); strides / offsets in bf16 elements
                                              // hidden_bf16: 1 = single bf16 operands (opt-in), 2 = split bf16 x3 (NSF sampler image)
  int oB_wk[SF_NBMAX], oB_w1[SF_NBMAX], oB_w2[SF_NBMAX];
  // 16-row-granular image of the incremental MAF inverse (v_mfma_f32_16x16x4_f32; sf_maf16.h) --------------
  //   tile = 16 samples x 16 rows in 4 VGPRs: lane l: sample l&15, rows 4*(l>>4)+r; degree groups packed whole
  //   into <= 4 tiles of 16 rows.  m16_ok = 0 when that packing does not exist (then the 32-row path is used).
  const float* packed16;
  int m16_ok, nT16, nC16, t16_stride;
  int t16_a;     // floats of part A of the 16-row image (everything but the fp32 hidden blocks): what k_maf_samp16 stages
  int t16_a_tab; // ... and of its prefix without the context block Wc: what it stages when the context table exists
  int o16_wh, o16_bh;  // head rows as one MFMA output tile (row 2q + ab of slot q; D <= 8, else -1) and their biases
  // fused first layer (round 5; D <= 8, else -1): W' = (W1 o M)(W0 o M0) as NT 16 x 16 fragments like o16_w0 -- there is no
  // activation between the initial layer and the first block's linear, so the two collapse into one product with the finished
  // dimensions; filled by k_maf_fuse16 from the packed image (not by the gather table), read by the fp32 sampler kernels
  // together with the table rows c0' = b1 + (W1 o M) c0
  int o16_wp;
  // split-bf16 hidden blocks of the 16-row sampler: 32-bit words (2 bf16 each), [ot][pair][hi|lo][64 lanes][4 words]
  const uint32_t* packed16B;
  int t16B_stride, nP16, o16B_wk[2];  // words per transform, in-tile pairs, block offsets in words
  int m16_span;  // 1: contiguous packing, degree groups may straddle tiles (g16_lo < g16_tile)
  int o16_w0, o16_wc, o16_b0, o16_wk[2], o16_bk[2], o16_hv, o16_hvb;
  int g16_tile[SF_DMAX];  // LAST tile that holds hidden units of MADE degree g ...
  int g16_lo[SF_DMAX];    // ... and the FIRST one (== g16_tile when every degree group sits inside one tile)
  // per-galaxy context table (sampling only; sf_flow_prepare_context): everything that depends on the context
  // row alone, evaluated once per galaxy instead of once per draw.  [gal][t][v][row], row in tile order:
  //   MAF (16-row path): v = 0: b0 + bc + Wc e(x)                                   R = nT16*16
  //   NSF              : v = 0: bin + Win_c e(x);  v = 1+k: bg_k + Wg_k e(x)         R = HT*32
  const float* ctab;
  int ctab_R, ctab_NV;
  // constants image ------------------------------------------------------------------------
  int c_pscale, c_pshift, c_tdim, c_xmean, c_xstd;  // tdim stored as float-encoded ints
  int c_dslot;  // MAF: [t][p-1] = physical slot of the dimension with MADE degree p (float-encoded)
  float logdet0;  // sum log|1/theta_std|
  float tail_bound, min_w, min_h, min_d, eps, lu_eps, inv_sqrt_h, deriv_const;
};

// pack-table flags of the 16-row MAF images (sf_layout.cpp; applied by k_pack / k_pack_bf16_split / k_train_prep):
//   second index of an fp32 entry == SF_PACK_TANH_SCALE: value x 2 log2(e)
//   bit 29 of a split-bf16 entry (SF_PACK_SPLIT_SCALED): the weight is multiplied by 2 log2(e) before it is split
#define SF_PACK_TANH_SCALE (-2)
#define SF_PACK_SPLIT_SCALED (1 << 29)
#define SF_PACK_SPLIT_INDEX 0x1fffffff
#define SF_TANH_PRESCALE 2.8853900817779268f

// ---- cooperative 16-row training image (sf_trainc.hip; MAF, num_blocks = 2, D <= 8, <= 4 hidden tiles of 16 rows) ----
// One workgroup = 8 waves = 64 samples; wave (grp, j) owns hidden tile j (grp 0) / NT-1-j (grp 1) of its group's 32
// samples, the waves exchange activation tiles through LDS at every layer.  Tile = 16 rows x 16 samples in 4 VGPRs
// (lane l: sample l & 15, rows 4*(l >> 4) + r), as in sf_maf16.hip.
//   input tile 0, row 4*g4 + r:  r < 2 and 2*g4 + r < D -> physical slot 2*g4 + r of u; every other row of tile 0 and
//                                all rows of tiles 1.. hold context features in increasing order (c_insrc, -1 = none)
//   head tile (one 16-row output tile), row 4*g4 + r: r < 2 -> a of slot 2*g4 + r, r >= 2 -> m of slot 2*g4 + r - 2:
//                                a lane's head outputs are exactly the (a, m) of the two slots it holds of u
//   forward block   [ot][it][64 lanes][4]: lane l, component r = W[ot*16 + (l & 15)][it*16 + 4*(l >> 4) + r]
//   transposed block[it][ot][64 lanes][4]: lane l, component r = W[ot*16 + 4*(l >> 4) + r][it*16 + (l & 15)]
//   gradient block  [ot][it][4 regs][64 lanes]: the accumulator of dW = delta . in^T as it stands
//                   (element (ro, ri) at (ro & 3) * 64 + (ro >> 2) * 16 + ri); bias gradients [tile*16 + row]
struct SfTrcDev {
  int ok;
  int NT, NI;                    // hidden tiles, input tiles
  int t_stride, g_stride;        // floats per transform: operand image / gradient partial
  int o_win, o_b0, o_w1, o_b1, o_w2, o_b2, o_wf, o_bf;   // forward blocks (offsets from the transform base)
  int o_wfT, o_w2T, o_w1T, o_winT;                       // transposed blocks
  int g_win, g_b0, g_w1, g_b1, g_w2, g_b2, g_wf, g_bf;   // gradient partial
  int kend[4];   // hidden output tile ot multiplies input tiles 0..kend[ot] (MADE masks; NT-1 = dense)
  int kbeg[4];   // hidden input tile it receives gradient from output tiles kbeg[it]..NT-1
  int c_insrc;   // constants image: [NI*16] float-encoded context feature of each input-tile row, -1 = none / slot row
  int c_jobs, n_jobs;  // constants image: [n_jobs] float-encoded (ot*4 + it) of the unmasked hidden weight blocks
};

// ---- cooperative 16-row NSF training image (sf_nsfc.hip; NSF, num_blocks = 2, 2 <= D <= 8, <= 5 hidden tiles: H <= 80) ----
// One workgroup = 4 (5 for five hidden tiles) waves = 32 samples (two 16-sample subtiles); wave j owns hidden tile j of both subtiles.  Tiles as above
// (lane l: sample l & 15, rows 4*(l >> 4) + r).  Rows are placed so that the k-components an MFMA step consumes (rows m,
// 4 + m, 8 + m, 12 + m of a tile for component m) fill up one after the other: a tile with n used rows costs
// ceil(n / 4) of the four v_mfma_f32_16x16x4_f32 steps.
//   hidden tile ot, row rho:            unit ot*16 + (rho >> 2) + 4*(rho & 3)                      (kc_h: components of the last tile)
//   input tile 0, row 4*g + m:          m < 2: theta dimension 2*g + m (zero weight where that dimension is transformed);
//                                       m >= 2: context feature (m - 2)*4 + g
//   input tile it >= 1, row 4*g + m:    context feature 8 + (it - 1)*16 + 4*m + g                  (kc_in[it] components)
//   spline-head tile j, row 4*g + r:    parameter slot 4*j + r of transformed dimension g (the lane that evaluates the spline
//                                       of (sample, dimension g) receives its own parameters); slots [0, KM) widths,
//                                       [KM, 2 KM) heights, [2 KM, 3 KM - 1) interior derivatives, KM = 8 (OTQ = 6) / 11 (OTQ = 8)
//   forward / transposed / gradient blocks exactly as SfTrcDev's
//   LU block: L[8][8] (strictly lower), U[8][8] (strictly upper), udiag[8], bias[8]; its gradient: block 0 = G . tt^T (dL below
//             the diagonal), block 1 = dt . u'^T (dU above it), then 16 row sums: [0, 8) dbias, [8, 16) d udiag
struct SfNscDev {
  int ok;
  int NT, NI, OTQ, KM;
  int kc_h, kc_in[3];
  int t_stride, g_stride;
  int o_win, o_bin, o_wg[2], o_bg[2], o_w1[2], o_b1[2], o_w2[2], o_b2[2], o_wout, o_bout, o_lu;   // forward blocks
  int o_woutT, o_w2T[2], o_w1T[2], o_winT;                                                        // transposed blocks
  int g_win, g_bin, g_wg[2], g_bg[2], g_w1[2], g_b1[2], g_w2[2], g_b2[2], g_wout, g_bout, g_lu;   // gradient partial
};

// NSF sampler image (sampling kernels only): the fp32 operand image WITHOUT the hidden W1 / W2 blocks, followed in LDS
// by those blocks as split bf16 -- hi = bf16(w), lo = bf16(w - hi), per block [mt][ks][hi | lo][64 lanes][8], the A operand of
// v_mfma_f32_32x32x16_bf16 in the k order of sf_bfrag -- so that hi.hi + hi.lo + lo.hi reproduce the fp32 product to ~2^-17
// at a fifth of the matrix-pipe time.  The tables live in SfLayout::src16a / src16b (fp32 part) and src16B (split part, logical
// index | part << 30), like the 16-row MAF sampler image.  log_prob and training never use it.
struct SfNsfSamp {
  int ok;
  int t_stride;   // floats per transform of the fp32 part
  int o_winu, o_winc, o_bin, o_wg[SF_NBMAX], o_bg[SF_NBMAX], o_b1[SF_NBMAX], o_b2[SF_NBMAX], o_wout, o_bout, o_lu;
  int tB_stride;  // bf16 elements per transform of the split part
  int oB_w1[SF_NBMAX], oB_w2[SF_NBMAX];
  // LDS staging plan (round 5): parts = whole items in execution order -- the input layer, block k (its fp32 pieces AND its
  // split W1 / W2), the spline head -- packed greedily into the 152 KiB budget; the production width H = 69 (three hidden
  // tiles) needs two parts, cfg3 one
  int n_parts, part_off[5], partB_off[5], blk_part[SF_NBMAX], head_part, part_floats_max, part_bytes_max;
};

struct SfLayout {
  SfDev dev;  // pointers left null
  SfNsfSamp nsfS;                     // nsfS.ok == 0: no NSF sampler image
  SfTrcDev trc;                       // trc.ok == 0: no cooperative training image for this flow (MAF)
  SfNscDev nsc;                       // nsc.ok == 0: ... (NSF); srcC1 / srcC2 / gdstC / n_imgC / n_gradC are shared with trc
  std::vector<int32_t> srcC1, srcC2;  // its gather table (sum of two sources, like src1/src2)
  std::vector<int32_t> gdstC;         // logical parameter -> index in a gradient partial (or -1)
  int64_t n_imgC = 0, n_gradC = 0;
  int64_t n_params = 0;
  int64_t n_packed = 0;  // floats in packed (== floats in packedT)
  std::vector<int32_t> src1, src2;    // forward image gather table
  std::vector<int32_t> srcT1, srcT2;  // transposed image gather table
  int64_t n_packedT = 0;
  // gradient image: same block offsets as the forward image, but inside a weight block the
  // order is [mt][kg][j][lane] (256 contiguous bytes per atomic wave-instruction) and a bias
  // block is [mt][32 rows].  gdst[i] = gradient-image index of logical parameter i, or -1.
  std::vector<int32_t> gdst;
  std::vector<int32_t> srcB;          // bf16 hidden image gather table (one entry per bf16 element)
  std::vector<int32_t> src16a, src16b;  // 16-row image gather table (sum of two sources, like src1/src2)
  std::vector<int32_t> src16B;          // split-bf16 hidden blocks of the 16-row sampler: per bf16 element, logical index |
                                        // (part << 30), part 0 = hi = bf16(w), 1 = lo = bf16(w - hi); -1 = zero
  int64_t n_packed16B = 0;
  int64_t n_packed16 = 0;
  int64_t n_packedB = 0;
  std::vector<float> cst;             // constants image
  std::string error;
};

static inline int sf_tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Builds tables; returns false and sets L.error on unsupported shapes.
bool sf_build_layout(const sf_flow_desc& d, SfLayout& L);

// ---- embedding MLP ---------------------------------------------------------------------------
#define SF_MLP_LMAX 4
struct SfMlpDev {
  const float* packed;
  const float* packedT;
  const float* cst;  // [x_mean n_in][x_std n_in]
  int n_in, n_out, L, act, HT;
  int nG[SF_MLP_LMAX];     // input groups of layer l
  int nGo[SF_MLP_LMAX];    // groups covering layer l's outputs (transposed operand K)
  int width[SF_MLP_LMAX];
  int o_w[SF_MLP_LMAX], o_b[SF_MLP_LMAX], oT_w[SF_MLP_LMAX];
};
struct SfMlpLayout {
  SfMlpDev dev;
  int64_t n_params = 0, n_packed = 0, n_packedT = 0;
  std::vector<int32_t> src1, src2, srcT1, srcT2, gdst;
  std::vector<float> cst;
  std::string error;
};
bool sf_build_mlp_layout(const sf_mlp_desc& d, SfMlpLayout& L);
