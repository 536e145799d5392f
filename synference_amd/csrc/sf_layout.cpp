// sf_layout.cpp -- host-side construction of the MFMA operand image tables.
// Pure host code (no HIP calls): exercised by the CPU test-suite through
// sf_flow_pack_table().  Logical layout: include/synference_hip.h.
#include "sf_layout.h"

#include <cmath>
#include <cstring>
#include <functional>

namespace {

struct Emitter {
  std::vector<int32_t>& s1;
  std::vector<int32_t>& s2;
  std::vector<int32_t>* gidx = nullptr;  // gradient-image index of each packed float
  int64_t cur = 0;  // write cursor (floats)

  int64_t pad_to(int64_t align) {
    while (cur % align) push(-1, -1);
    return cur;
  }
  void push(int32_t a, int32_t b, int64_t g = -1) {
    s1.push_back(a);
    s2.push_back(b);
    if (gidx) gidx->push_back((int32_t)(g >= 0 ? g : cur));
    ++cur;
  }
  // Weight block: OT output tiles x nG input groups x 64 lanes x 4.
  // logical index of element (o,i) = base + o*so + i*si ; mask(o,i) false -> structural zero.
  int64_t linear(int OT, int nG, const std::vector<int>& orow, const std::vector<int>& irow,
                 int64_t base, int64_t so, int64_t si,
                 const std::function<bool(int, int)>& mask) {
    int64_t start = cur;
    for (int mt = 0; mt < OT; ++mt)
      for (int kg = 0; kg < nG; ++kg)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 4; ++j) {
            int o = orow[mt * 32 + (l & 31)];
            int i = irow[kg * 8 + 4 * (l >> 5) + j];
            const int64_t g = start + (((int64_t)mt * nG + kg) * 4 + j) * 64 + l;
            if (o >= 0 && i >= 0 && (!mask || mask(o, i)))
              push((int32_t)(base + o * so + i * si), -1, g);
            else
              push(-1, -1, g);
          }
    return start;
  }
  // Bias block [OT][2][16]; up to two logical sources summed (b0 + bc).
  int64_t bias(int OT, const std::vector<int>& orow, int64_t base1, int64_t base2) {
    int64_t start = cur;
    for (int mt = 0; mt < OT; ++mt)
      for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
          int o = orow[mt * 32 + sf_tile_row(r, h)];
          const int64_t g = start + mt * 32 + sf_tile_row(r, h);
          if (o >= 0)
            push((int32_t)(base1 + o), base2 >= 0 ? (int32_t)(base2 + o) : -1, g);
          else
            push(-1, -1, g);
        }
    return start;
  }
};

std::vector<int> iota_rows(int n_valid, int n_total) {
  std::vector<int> v(n_total, -1);
  for (int i = 0; i < n_valid && i < n_total; ++i) v[i] = i;
  return v;
}

int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace

bool sf_build_layout(const sf_flow_desc& d, SfLayout& L) {
  auto fail = [&](const std::string& m) {
    L.error = m;
    return false;
  };
  if (d.kind != SF_MAF && d.kind != SF_NSF) return fail("kind must be SF_MAF or SF_NSF");
  if (d.D < 1 || d.D > SF_DMAX) return fail("D must be in 1..16");
  if (d.C < 1 || d.C > 512) return fail("C must be in 1..512");
  if (d.H < 1 || d.H > 128) return fail("H must be in 1..128");
  if (d.T < 1 || d.T > 64) return fail("T must be in 1..64");
  if (d.NB < 1 || d.NB > SF_NBMAX) return fail("NB must be in 1..4");
  if (d.kind == SF_NSF && (d.K < 2 || d.K > 16)) return fail("K must be in 2..16");
  if (!d.theta_mean || !d.theta_std || !d.x_mean || !d.x_std)
    return fail("z-score buffers must be given");

  const int D = d.D, C = d.C, H = d.H, T = d.T, NB = d.NB, K = d.K;
  SfDev& v = L.dev;
  std::memset(&v, 0, sizeof(v));
  std::memset(&L.trc, 0, sizeof(L.trc));
  std::memset(&L.nsc, 0, sizeof(L.nsc));
  std::memset(&L.nsfS, 0, sizeof(L.nsfS));
  v.kind = d.kind; v.D = D; v.C = C; v.H = H; v.T = T; v.K = K; v.NB = NB;
  v.scale_fn = d.scale_fn;
  v.HT = ceil_div(H, 32);
  v.nGu = ceil_div(D, 8);
  v.nGc = ceil_div(C, 8);
  v.nGh = ceil_div(H, 8);
  v.tail_bound = d.tail_bound; v.min_w = d.min_bin_width; v.min_h = d.min_bin_height;
  v.min_d = d.min_derivative; v.eps = d.maf_eps; v.lu_eps = d.lu_eps;
  v.inv_sqrt_h = (float)(1.0 / std::sqrt((double)H));
  v.deriv_const = (float)std::log(std::exp(1.0 - (double)d.min_derivative) - 1.0);

  // ---- hidden physical row -> logical unit ---------------------------------------------------
  // NSF: identity.  MAF: units sorted by MADE degree, every degree group kept inside one tile when
  // that fits in <= 4 tiles (then the masked HxH blocks are block-lower-triangular in tile/group
  // units and the autoregressive inverse can be evaluated incrementally).
  std::vector<int> hrow_full(4 * 32, -1);
  std::vector<int> h16row(4 * 16, -1);
  v.inc_ok = 0;
  v.m16_ok = 0;
  if (d.kind == SF_MAF) {
    const int mx = std::max(1, D - 1), mn = std::min(1, D - 1);
    const int G = mx;  // degree values mn .. mn+G-1
    std::vector<std::vector<int>> grp(G);
    for (int j = 0; j < H; ++j) grp[j % mx].push_back(j);  // degree = j % mx + mn
    // aligned placement
    std::vector<int> rows(4 * 32, -1);
    int tile = 0, used = 0;
    bool ok = true;
    std::vector<int> gt(G), gend(G);
    for (int g = 0; g < G && ok; ++g) {
      const int n = (int)grp[g].size();
      if (n > 32) { ok = false; break; }
      if (used + n > 32) { ++tile; used = 0; }
      if (tile >= 4) { ok = false; break; }
      for (int q = 0; q < n; ++q) rows[tile * 32 + used + q] = grp[g][q];
      used += n;
      gt[g] = tile;
      gend[g] = tile * 32 + used;  // exclusive end row of degrees <= this one
    }
    // Aligned placement (one pass = one tile) costs extra tiles when the groups pack badly (H = 64, D = 8: 3 tiles
    // instead of 2, i.e. 1.5x the MFMA work of log_prob and training) and extra input groups, which can push the
    // transform's image out of the LDS.  Contiguous placement (a group may straddle tiles, its pass recomputes
    // every tile it touches) is taken instead when it saves a tile or is the only one of the two that fits the LDS.
    if (ok && D >= 2) {
      const int nGu_e = ceil_div(D, 8), nGc_e = ceil_div(C, 8);
      auto image_floats = [&](int ht, int ngh) {
        return 256L * (ht * (nGu_e + nGc_e) + (long)NB * ht * ngh + ngh) + 64L * D * ht + 64L * (NB + 1) * ht;
      };
      const long lds_floats = 152L * 1024 / 4;
      const int ht_a = tile + 1, ngh_a = ceil_div(gend[G - 1], 8), ht_c = ceil_div(H, 32), ngh_c = ceil_div(H, 8);
      if (ht_a > ht_c || (image_floats(ht_a, ngh_a) > lds_floats && image_floats(ht_c, ngh_c) <= lds_floats)) ok = false;
    }
    if (ok && D <= SF_DMAX) {
      hrow_full = rows;
      v.HT = std::max(v.HT, tile + 1);
      v.inc_ok = (D >= 2) ? 1 : 0;
      for (int g = 0; g < G; ++g) {
        // degree value (g + mn) is stored at index (g + mn); index 0 unused when mn == 1
        v.g_tile[g + mn] = gt[g];
        v.g_lo[g + mn] = gt[g];
        v.g_kend[g + mn] = ceil_div(gend[g], 8);
      }
      v.nGh = ceil_div(gend[G - 1], 8);
    } else {  // degree-sorted, contiguous rows: a degree group may straddle tiles (g_lo < g_tile)
      int r = 0;
      for (int g = 0; g < G; ++g) {
        v.g_lo[g + mn] = r / 32;
        for (int j : grp[g]) hrow_full[r++] = j;
        v.g_tile[g + mn] = (r - 1) / 32;
        v.g_kend[g + mn] = ceil_div(r, 8);
      }
      v.inc_ok = (D >= 2 && D <= SF_DMAX && H <= 128) ? 1 : 0;
    }
    // 16-row tiles for the 16x16x4 incremental inverse: whole degree groups, at most 4 tiles
    {
      h16row.assign(4 * 16, -1);
      int tile16 = 0, used16 = 0;
      bool ok16 = (D >= 2 && NB <= 2);
      for (int g = 0; g < G && ok16; ++g) {
        const int n = (int)grp[g].size();
        if (n > 16) { ok16 = false; break; }
        if (used16 + n > 16) { ++tile16; used16 = 0; }
        if (tile16 >= 4) { ok16 = false; break; }
        for (int q = 0; q < n; ++q) h16row[tile16 * 16 + used16 + q] = grp[g][q];
        used16 += n;
        v.g16_tile[g + mn] = tile16;
      }
      for (int g = 0; g < G; ++g) v.g16_lo[g + mn] = v.g16_tile[g + mn];
      if (!ok16 && D >= 2 && NB <= 2 && H <= 64) {
        // whole groups do not fit four tiles: pack the degree-sorted units contiguously; a group may then straddle
        // tiles and its pass recomputes every tile it touches (sf_pass16_span)
        h16row.assign(4 * 16, -1);
        int r = 0;
        for (int g = 0; g < G; ++g) {
          v.g16_lo[g + mn] = r / 16;
          for (int j : grp[g]) h16row[r++] = j;
          v.g16_tile[g + mn] = (r - 1) / 16;
        }
        tile16 = (H - 1) / 16;
        ok16 = true;
        v.m16_span = 1;
      }
      v.m16_ok = ok16 ? 1 : 0;
      v.nT16 = ok16 ? tile16 + 1 : 0;
      v.nC16 = ceil_div(C, 16);
    }
  } else {
    for (int j = 0; j < H; ++j) hrow_full[j] = j;
  }
  const int HT = v.HT;
  for (int mt = 0; mt < 4; ++mt) v.mt_kend[mt] = v.nGh;
  if (d.kind == SF_MAF && v.inc_ok) {
    const int mn = std::min(1, D - 1), G = std::max(1, D - 1);
    for (int mt = 0; mt < 4; ++mt) {
      int k = 0;
      for (int g = 0; g < G; ++g)
        if (v.g_lo[g + mn] <= mt && mt <= v.g_tile[g + mn]) k = std::max(k, v.g_kend[g + mn]);  // groups with rows in mt
      if (k > 0) v.mt_kend[mt] = k;
    }
  } else if (d.kind == SF_MAF && D >= 2 && H <= 128) {
    // contiguous degree-sorted rows: output tile mt holds degrees <= that of its last row and needs the input
    // groups that cover all units of those degrees (block-triangular skipping still applies)
    const int mx = std::max(1, D - 1);
    std::vector<int> cnt(mx, 0);
    for (int j = 0; j < H; ++j) ++cnt[j % mx];
    std::vector<int> endrow(mx, 0);
    for (int g = 0, r = 0; g < mx; ++g) { r += cnt[g]; endrow[g] = r; }
    for (int mt = 0; mt < v.HT; ++mt) {
      const int last = std::min(H, (mt + 1) * 32) - 1;   // last occupied row of the tile
      int g = 0;
      while (g < mx - 1 && endrow[g] <= last) ++g;        // degree group of that row
      v.mt_kend[mt] = std::min(v.nGh, ceil_div(endrow[g], 8));
    }
  }
  const std::vector<int> hrow_out(hrow_full.begin(), hrow_full.begin() + HT * 32);
  const std::vector<int> hrow_in(hrow_full.begin(), hrow_full.begin() + v.nGh * 8);
  const std::vector<int> crow_in = iota_rows(C, v.nGc * 8);
  v.hidden_bf16 = d.hidden_bf16 ? 1 : 0;
  v.nKS = ceil_div(v.nGh, 2);
  // bf16 hidden block: [mt][ks][lane][8]; element j of lane (c,h) is input row 16ks + 8(j>>2) + 4h + (j&3)
  // (the k-order in which v_mfma_f32_32x32x16_bf16 consumes registers 8ks'..8ks'+7 of an accumulator tile)
  auto emit_bf16 = [&](int64_t base, const std::function<bool(int, int)>& mask) -> int64_t {
    while (L.srcB.size() % 8) L.srcB.push_back(-1);
    const int64_t start = (int64_t)L.srcB.size();
    for (int mt = 0; mt < HT; ++mt)
      for (int ks = 0; ks < v.nKS; ++ks)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int o = hrow_full[mt * 32 + (l & 31)];
            const int rho = 16 * ks + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
            const int i = rho < 128 ? hrow_full[rho] : -1;
            L.srcB.push_back((o >= 0 && i >= 0 && (!mask || mask(o, i))) ? (int32_t)(base + (int64_t)o * H + i) : -1);
          }
    return start;
  };

  // ---- MAF physical slot maps: sigma_T = identity, sigma_t = sigma_{t+1} o perm_t^{-1} ----
  std::vector<std::vector<int>> sigma(T + 1, std::vector<int>(D));
  for (int i = 0; i < D; ++i) sigma[T][i] = i;
  if (d.kind == SF_MAF) {
    for (int t = T - 1; t >= 0; --t) {
      std::vector<int> perm(D);
      for (int j = 0; j < D; ++j) perm[j] = d.perms ? d.perms[t * D + j] : j;
      std::vector<char> seen(D, 0);
      for (int j = 0; j < D; ++j) {
        if (perm[j] < 0 || perm[j] >= D || seen[perm[j]]) return fail("perms is not a permutation");
        seen[perm[j]] = 1;
      }
      // logical j of transform t+1 input  <-  logical perm[j] of transform t output
      for (int j = 0; j < D; ++j) sigma[t][perm[j]] = sigma[t + 1][j];
    }
  } else {
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < D; ++i) sigma[t][i] = i;
  }

  // ---- constants image ---------------------------------------------------------------------
  {
    std::vector<int> tdim(D);  // phys slot p holds logical theta dim tdim[p]
    for (int i = 0; i < D; ++i) tdim[sigma[0][i]] = i;
    v.c_pscale = 0; v.c_pshift = SF_DMAX; v.c_tdim = 2 * SF_DMAX;
    v.c_xmean = 3 * SF_DMAX; v.c_xstd = 3 * SF_DMAX + ((C + 3) / 4) * 4;
    L.cst.assign(v.c_xstd + ((C + 3) / 4) * 4, 0.f);
    double ld0 = 0;
    for (int p = 0; p < D; ++p) {
      float sd = d.theta_std[tdim[p]], mu = d.theta_mean[tdim[p]];
      float sc = 1.0f / sd;
      L.cst[v.c_pscale + p] = sc;
      L.cst[v.c_pshift + p] = -mu / sd;
      L.cst[v.c_tdim + p] = (float)tdim[p];
      ld0 += std::log(std::fabs((double)sc));
    }
    v.logdet0 = (float)ld0;
    for (int i = 0; i < C; ++i) {
      L.cst[v.c_xmean + i] = d.x_mean[i];
      L.cst[v.c_xstd + i] = d.x_std[i];
    }
    for (int i = C; i < ((C + 3) / 4) * 4; ++i) L.cst[v.c_xstd + i] = 1.f;
    v.c_dslot = (int)L.cst.size();
    L.cst.resize(L.cst.size() + (size_t)T * SF_DMAX, 0.f);
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < D; ++i) L.cst[v.c_dslot + t * SF_DMAX + i] = (float)sigma[t][i];  // degree i+1
  }

  std::vector<int32_t> gidx;
  Emitter E{L.src1, L.src2, &gidx};
  Emitter ET{L.srcT1, L.srcT2, nullptr};  // transposed image
  int64_t P = 0;  // logical cursor

  if (d.kind == SF_MAF) {
    const int mx = std::max(1, D - 1), mn = std::min(1, D - 1);
    auto deg_h = [&](int j) { return j % mx + mn; };
    for (int t = 0; t < T; ++t) {
      const int64_t tb = E.pad_to(64);
      if (t == 1) v.t_stride = (int)tb;
      // logical offsets
      const int64_t lW0 = P, lb0 = lW0 + (int64_t)H * D, lWc = lb0 + H, lbc = lWc + (int64_t)H * C;
      int64_t cur = lbc + H;
      int64_t lWk[SF_NBMAX], lbk[SF_NBMAX];
      for (int k = 0; k < NB; ++k) { lWk[k] = cur; lbk[k] = cur + (int64_t)H * H; cur = lbk[k] + H; }
      const int64_t lWf = cur, lbf = lWf + (int64_t)2 * D * H;
      P = lbf + 2 * D;

      std::vector<int> sinv(D);  // phys slot -> logical dim of this transform
      for (int i = 0; i < D; ++i) sinv[sigma[t][i]] = i;
      std::vector<int> urow(v.nGu * 8, -1);
      for (int p = 0; p < D; ++p) urow[p] = sinv[p];
      std::vector<int> frow(32, -1);
      for (int p = 0; p < D; ++p) {
        frow[sf_tile_row(2 * (p >> 1), p & 1)] = 2 * sinv[p];
        frow[sf_tile_row(2 * (p >> 1) + 1, p & 1)] = 2 * sinv[p] + 1;
      }
      int o;
      o = (int)(E.linear(HT, v.nGu, hrow_out, urow, lW0, D, 1,
                         [&](int j, int i) { return deg_h(j) >= i + 1; }) - tb);
      if (t == 0) v.o_w0 = o;
      o = (int)(E.linear(HT, v.nGc, hrow_out, crow_in, lWc, C, 1, nullptr) - tb);
      if (t == 0) v.o_wc = o;
      o = (int)(E.bias(HT, hrow_out, lb0, lbc) - tb);
      if (t == 0) v.o_b0 = o;
      for (int k = 0; k < NB; ++k) {
        o = (int)(E.linear(HT, v.nGh, hrow_out, hrow_in, lWk[k], H, 1,
                           [&](int j, int i) { return deg_h(j) >= deg_h(i); }) - tb);
        if (t == 0) v.o_wk[k] = o;
        o = (int)(E.bias(HT, hrow_out, lbk[k], -1) - tb);
        if (t == 0) v.o_bk[k] = o;
        if (v.hidden_bf16) {
          if (k == 0) {
            while (L.srcB.size() % 64) L.srcB.push_back(-1);
            if (t == 1) v.tB_stride = (int)L.srcB.size();
          }
          const int64_t tbB = (int64_t)t * (t >= 1 ? v.tB_stride : 0);
          const int64_t sB = emit_bf16(lWk[k], [&](int j, int i) { return deg_h(j) >= deg_h(i); });
          if (t == 0) v.oB_wk[k] = (int)(sB - tbB);
        }
      }
      o = (int)(E.linear(1, v.nGh, frow, hrow_in, lWf, H, 1,
                         [&](int oo, int i) { return (oo / 2 + 1) > deg_h(i); }) - tb);
      if (t == 0) v.o_wf = o;
      o = (int)(E.bias(1, frow, lbf, -1) - tb);
      if (t == 0) v.o_bf = o;
      // head rows once more, laid out for a per-lane dot product (incremental inverse): for physical
      // slot q the (a, m) rows, split by row half h, indexed like the lane's activation registers
      {
        int64_t s0 = E.pad_to(4);
        if (t == 0) v.o_hv = (int)(s0 - tb);
        for (int q = 0; q < D; ++q)
          for (int ab = 0; ab < 2; ++ab)
            for (int h = 0; h < 2; ++h)
              for (int mt = 0; mt < HT; ++mt)
                for (int r = 0; r < 16; ++r) {
                  const int unit = hrow_full[mt * 32 + sf_tile_row(r, h)];
                  const int orow_l = 2 * sinv[q] + ab;
                  const bool on = unit >= 0 && (orow_l / 2 + 1) > deg_h(unit);
                  E.push(on ? (int32_t)(lWf + (int64_t)orow_l * H + unit) : -1, -1);
                }
        int64_t s1 = E.pad_to(4);
        if (t == 0) v.o_hvb = (int)(s1 - tb);
        for (int q = 0; q < D; ++q)
          for (int ab = 0; ab < 2; ++ab) E.push((int32_t)(lbf + 2 * sinv[q] + ab), -1);
      }
      // ---- transposed operands for the data-gradient pass: delta_in = W^T delta_out ----
      {
        const int64_t tbT = ET.pad_to(64);
        if (t == 1) v.tT_stride = (int)tbT;
        v.nGf = ceil_div(2 * ((D + 1) / 2), 4);
        std::vector<int> urow32(32, -1);
        for (int p = 0; p < D; ++p) urow32[p] = sinv[p];
        // "o" = forward INPUT index, "i" = forward OUTPUT index: W[i][o] at base + o + i*in_dim
        o = (int)(ET.linear(HT, v.nGf, hrow_out, frow, lWf, 1, H,
                            [&](int unit, int oo) { return (oo / 2 + 1) > deg_h(unit); }) - tbT);
        if (t == 0) v.oT_wf = o;
        for (int k = 0; k < NB; ++k) {
          o = (int)(ET.linear(HT, v.nGh, hrow_out, hrow_in, lWk[k], 1, H,
                              [&](int uin, int uout) { return deg_h(uout) >= deg_h(uin); }) - tbT);
          if (t == 0) v.oT_wk[k] = o;
        }
        o = (int)(ET.linear(1, v.nGh, urow32, hrow_in, lW0, 1, D,
                            [&](int dim, int unit) { return deg_h(unit) >= dim + 1; }) - tbT);
        if (t == 0) v.oT_w0 = o;
        const int CT = ceil_div(C, 32);
        o = (int)(ET.linear(CT, v.nGh, iota_rows(C, CT * 32), hrow_in, lWc, 1, C, nullptr) - tbT);
        if (t == 0) v.oT_wc = o;
      }
      // ---- 16-row-granular image for the incremental inverse ------------------------------------
      if (v.m16_ok) {
        auto push16 = [&](int32_t a, int32_t b) { L.src16a.push_back(a); L.src16b.push_back(b); };
        while (L.src16a.size() % 1024) push16(-1, -1);  // whole 4 KiB groups: the staging loop copies 4 x 1 KiB per address
        const int64_t tb16 = (int64_t)L.src16a.size();
        if (t == 1) v.t16_stride = (int)tb16;
        auto here = [&]() { return (int)((int64_t)L.src16a.size() - tb16); };
        // weight block [ot][it][64 lanes][4]: lane l: out row ot*16+(l&15), in rows it*16 + 4*(l>>4) + r
        // second index SF_PACK_TANH_SCALE (-2): the packed value is multiplied by 2 log2(e) -- the hidden blocks' weights and
        // biases of the 16-row images carry the factor of tanh(x) = 1 - 2 / (1 + 2^(2 log2(e) x)), so that the kernels'
        // tanh starts at the exp2 (the kernel is bound by vector issue: one multiply per activation less)
        auto linear16 = [&](int OT, int IT, const std::vector<int>& orow, const std::vector<int>& irow, int64_t base,
                            int in_dim, const std::function<bool(int, int)>& mask, int32_t second = -1) {
          for (int ot = 0; ot < OT; ++ot)
            for (int it = 0; it < IT; ++it)
              for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                  const int oo = orow[ot * 16 + (l & 15)], ii = irow[it * 16 + 4 * (l >> 4) + r];
                  const bool on = oo >= 0 && ii >= 0 && (!mask || mask(oo, ii));
                  push16(on ? (int32_t)(base + (int64_t)oo * in_dim + ii) : -1, on ? second : -1);
                }
        };
        auto bias16 = [&](int OT, const std::vector<int>& orow, int64_t b1, int64_t b2) {
          for (int ot = 0; ot < OT; ++ot)
            for (int g4 = 0; g4 < 4; ++g4)
              for (int r = 0; r < 4; ++r) {
                const int oo = orow[ot * 16 + 4 * g4 + r];
                push16(oo >= 0 ? (int32_t)(b1 + oo) : -1,
                       oo < 0 ? -1 : (b2 >= 0 ? (int32_t)(b2 + oo) : (b2 == SF_PACK_TANH_SCALE ? SF_PACK_TANH_SCALE : -1)));
              }
        };
        std::vector<int> u16(16, -1);
        for (int p = 0; p < D; ++p) u16[p] = sinv[p];
        const std::vector<int> c16 = iota_rows(C, v.nC16 * 16);
        const int NT = v.nT16;
        // part A (everything but the hidden H x H blocks; staged by both sampler kernels)
        if (t == 0) v.o16_w0 = here();
        linear16(NT, 1, h16row, u16, lW0, D, [&](int j, int i) { return deg_h(j) >= i + 1; });
        if (t == 0) v.o16_b0 = here();
        bias16(NT, h16row, lb0, lbc);
        for (int k = 0; k < NB && k < 2; ++k) {
          if (t == 0) v.o16_bk[k] = here();
          bias16(NT, h16row, lbk[k], SF_PACK_TANH_SCALE);
        }
        // head rows for the per-lane dot product: [slot][g4][tile (4)][r][a|m]  (a and m interleaved: one packed
        // v_pk_fma_f32 updates both partial sums)
        if (t == 0) v.o16_hv = here();
        for (int q = 0; q < D; ++q)
          for (int g4 = 0; g4 < 4; ++g4)
            for (int tl = 0; tl < 4; ++tl)
              for (int r = 0; r < 4; ++r)
                for (int ab = 0; ab < 2; ++ab) {
                  const int unit = tl < NT ? h16row[tl * 16 + 4 * g4 + r] : -1;
                  const int orow_l = 2 * sinv[q] + ab;
                  const bool on = unit >= 0 && (orow_l / 2 + 1) > deg_h(unit);
                  push16(on ? (int32_t)(lWf + (int64_t)orow_l * H + unit) : -1, -1);
                }
        if (t == 0) v.o16_hvb = here();
        for (int q = 0; q < D; ++q)
          for (int ab = 0; ab < 2; ++ab) push16((int32_t)(lbf + 2 * sinv[q] + ab), -1);
        // the same head rows as ONE 16-row output tile of the matrix pipe (D <= 8): row 2q + ab = (a | m) of physical
        // slot q, one A fragment per hidden tile, and the head biases as that tile's initial accumulator
        if (D <= 8) {
          while (L.src16a.size() % 4) push16(-1, -1);
          if (t == 0) v.o16_wh = here();
          std::vector<int> hrow16(16, -1);
          for (int q = 0; q < D; ++q)
            for (int ab = 0; ab < 2; ++ab) hrow16[2 * q + ab] = 2 * sinv[q] + ab;
          linear16(1, NT, hrow16, h16row, lWf, H, [&](int oo, int unit) { return (oo / 2 + 1) > deg_h(unit); });
          if (t == 0) v.o16_bh = here();
          bias16(1, hrow16, lbf, -1);
          while (L.src16a.size() % 4) push16(-1, -1);
          if (t == 0) v.o16_wp = here();
          for (int i = 0; i < NT * 256; ++i) push16(-1, -1);   // (computed on the device: k_maf_fuse16)
        } else if (t == 0) {
          v.o16_wh = v.o16_bh = -1;
          v.o16_wp = -1;
        }
        // what the sampler stages when the per-galaxy context table exists (it then never reads Wc) ends here
        while (L.src16a.size() % 1024) push16(-1, -1);
        if (t == 0) v.t16_a_tab = here();
        if (t == 0) v.o16_wc = here();
        linear16(NT, v.nC16, h16row, c16, lWc, C, nullptr);
        while (L.src16a.size() % 1024) push16(-1, -1);
        if (t == 0) v.t16_a = here();
        // part B: the hidden blocks in fp32 (k_maf_inv16: parity hook, acceptance counts, explicit rounds)
        for (int k = 0; k < NB && k < 2; ++k) {
          if (t == 0) v.o16_wk[k] = here();
          linear16(NT, NT, h16row, h16row, lWk[k], H, [&](int j, int i) { return deg_h(j) >= deg_h(i); }, SF_PACK_TANH_SCALE);
        }
        // ---- split-bf16 image of the hidden blocks (k_maf_samp16): per block [ot][pair][hi|lo][64 lanes][8 bf16];
        // element j of lane l is W[out row ot*16 + (l&15)][in row 16*(2*pair + (j>>2)) + 4*(l>>4) + (j&3)] -- the k order
        // in which two 16-row activation tiles (4 registers each, lane = sample + 16 * row group) form the B operand
        // of v_mfma_f32_16x16x32_bf16 without moving between lanes.  hi = bf16(w), lo = bf16(w - hi): three products
        // hi.hi + hi.lo + lo.hi reproduce the fp32 product to ~2^-17 relative.
        {
          while (L.src16B.size() % 2048) L.src16B.push_back(-1);  // whole 4 KiB groups (2 bf16 per 32-bit word)
          const int64_t tbB = (int64_t)L.src16B.size();
          if (t == 1) v.t16B_stride = (int)(tbB / 2);
          const int NP = (NT + 1) / 2;
          v.nP16 = NP;
          for (int k = 0; k < NB && k < 2; ++k) {
            if (t == 0) v.o16B_wk[k] = (int)(((int64_t)L.src16B.size() - tbB) / 2);
            for (int ot = 0; ot < NT; ++ot)
              for (int pr = 0; pr < NP; ++pr) {
                // aligned placement: tiles above `ot` hold strictly higher degrees, so pairs above ot's own are all
                // masked and are not stored (entry index ot + (ot == 3) + pr); 36 KB instead of 44 KB per transform for
                // four tiles = a fourth workgroup per CU.  Contiguous placement keeps every pair.
                if (!v.m16_span && pr > ot / 2) continue;
                for (int part = 0; part < 2; ++part)
                  for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                      const int it = 2 * pr + (j >> 2);
                      const int oo = h16row[ot * 16 + (l & 15)];
                      const int ii = it < NT ? h16row[it * 16 + 4 * (l >> 4) + (j & 3)] : -1;
                      const bool on = oo >= 0 && ii >= 0 && deg_h(oo) >= deg_h(ii);
                      L.src16B.push_back(on ? (int32_t)((lWk[k] + (int64_t)oo * H + ii) | ((int64_t)part << 30) | SF_PACK_SPLIT_SCALED) : -1);
                    }
              }
          }
        }
      }
      // ---- cooperative 16-row training image (sf_layout.h, SfTrcDev; sf_trainc.hip) -------------------------
      if (v.m16_ok && NB == 2 && D <= 8) {
        SfTrcDev& c = L.trc;
        const int NT = v.nT16;
        if (t == 0) {
          c.ok = 1;
          c.NT = NT;
          // rows of the input tiles: slot rows first (fixed), context features fill what is left
          std::vector<int> insrc(16, -1);
          int f = 0;
          for (int rho = 0; rho < 16 && f < C; ++rho) {
            const bool slot_row = (rho & 3) < 2 && 2 * (rho >> 2) + (rho & 3) < D;
            if (!slot_row) insrc[rho] = f++;
          }
          while (f < C) {
            for (int rho = 0; rho < 16; ++rho) insrc.push_back(f < C ? f++ : -1);
          }
          c.NI = (int)insrc.size() / 16;
          c.c_insrc = (int)L.cst.size();
          for (int q : insrc) L.cst.push_back((float)q);
          // unmasked hidden blocks and the tile bounds they imply
          for (int ot = 0; ot < 4; ++ot) { c.kend[ot] = 0; c.kbeg[ot] = NT > 0 ? NT - 1 : 0; }
          c.c_jobs = (int)L.cst.size();
          c.n_jobs = 0;
          for (int ot = 0; ot < NT; ++ot)
            for (int it = 0; it < NT; ++it) {
              bool any = false;
              for (int ro = 0; ro < 16 && !any; ++ro)
                for (int ri = 0; ri < 16 && !any; ++ri) {
                  const int oo = h16row[ot * 16 + ro], ii = h16row[it * 16 + ri];
                  any = oo >= 0 && ii >= 0 && deg_h(oo) >= deg_h(ii);
                }
              if (any) {
                c.kend[ot] = std::max(c.kend[ot], it);
                c.kbeg[it] = std::min(c.kbeg[it], ot);
                L.cst.push_back((float)(ot * 4 + it));
                ++c.n_jobs;
              }
            }
          while (L.cst.size() % 4) L.cst.push_back(0.f);
        }
        const int NI = c.NI;
        auto insrc_at = [&](int it, int rho) { return (int)L.cst[c.c_insrc + it * 16 + rho]; };
        auto pushC = [&](int32_t a, int32_t b) { L.srcC1.push_back(a); L.srcC2.push_back(b); };
        while (L.srcC1.size() % 256) pushC(-1, -1);
        const int64_t tbC = (int64_t)L.srcC1.size();
        if (t == 1) c.t_stride = (int)tbC;
        auto hereC = [&]() { return (int)((int64_t)L.srcC1.size() - tbC); };
        int64_t gcur = 0;                                  // cursor in this transform's gradient partial
        const int64_t gbase_t = (int64_t)t * c.g_stride;   // (g_stride is known from t = 0 on)
        // logical index of W[(to, ro)][(ti, ri)] for every layer, -1 = structural zero
        auto w_in = [&](int to, int ro, int ti, int ri) -> int64_t {
          const int oo = h16row[to * 16 + ro];
          if (oo < 0) return -1;
          if (ti == 0 && (ri & 3) < 2) {
            const int p = 2 * (ri >> 2) + (ri & 3);
            if (p < D) return deg_h(oo) >= sinv[p] + 1 ? lW0 + (int64_t)oo * D + sinv[p] : -1;
          }
          const int fsrc = insrc_at(ti, ri);
          return fsrc >= 0 ? lWc + (int64_t)oo * C + fsrc : -1;
        };
        auto w_hid = [&](int64_t base) {
          return [&, base](int to, int ro, int ti, int ri) -> int64_t {
            const int oo = h16row[to * 16 + ro], ii = h16row[ti * 16 + ri];
            return (oo >= 0 && ii >= 0 && deg_h(oo) >= deg_h(ii)) ? base + (int64_t)oo * H + ii : -1;
          };
        };
        auto head_row = [&](int ro) -> int {  // logical output row of head-tile row ro, -1 = padding
          const int p = 2 * (ro >> 2) + (ro & 1);
          if (p >= D) return -1;
          return 2 * sinv[p] + ((ro & 3) >= 2 ? 1 : 0);
        };
        auto w_head = [&](int /*to*/, int ro, int ti, int ri) -> int64_t {
          const int orow_l = head_row(ro), ii = h16row[ti * 16 + ri];
          return (orow_l >= 0 && ii >= 0 && (orow_l / 2 + 1) > deg_h(ii)) ? lWf + (int64_t)orow_l * H + ii : -1;
        };
        using WFn = std::function<int64_t(int, int, int, int)>;
        auto fwd_block = [&](int OT, int IT, const WFn& fn, int64_t g_off) {
          for (int ot = 0; ot < OT; ++ot)
            for (int it = 0; it < IT; ++it)
              for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                  const int ro = l & 15, ri = 4 * (l >> 4) + r;
                  const int64_t idx = fn(ot, ro, it, ri);
                  pushC((int32_t)idx, -1);
                  if (idx >= 0) L.gdstC[(size_t)idx] = (int32_t)(gbase_t + g_off + ((int64_t)ot * IT + it) * 256 + (ro & 3) * 64 + (ro >> 2) * 16 + ri);
                }
        };
        auto tr_block = [&](int ITr, int OTk, const WFn& fn) {  // rows = forward input tiles, K = forward output tiles
          for (int it = 0; it < ITr; ++it)
            for (int ot = 0; ot < OTk; ++ot)
              for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) pushC((int32_t)fn(ot, 4 * (l >> 4) + r, it, l & 15), -1);
        };
        auto bias_block = [&](int OT, const std::function<int64_t(int, int)>& f1, const std::function<int64_t(int, int)>& f2,
                              int64_t g_off) {
          for (int ot = 0; ot < OT; ++ot)
            for (int ro = 0; ro < 16; ++ro) {
              const int64_t a1 = f1(ot, ro), a2 = f2 ? f2(ot, ro) : -1;
              pushC((int32_t)a1, (int32_t)a2);
              const int32_t g = (int32_t)(gbase_t + g_off + ot * 16 + ro);
              if (a1 >= 0) L.gdstC[(size_t)a1] = g;
              if (a2 >= 0) L.gdstC[(size_t)a2] = g;
            }
        };
        auto hid_bias = [&](int64_t base) {
          return [&, base](int ot, int ro) -> int64_t { const int oo = h16row[ot * 16 + ro]; return oo >= 0 ? base + oo : -1; };
        };
        if (t == 0) L.gdstC.assign((size_t)0, -1);
        if (L.gdstC.size() < (size_t)P) L.gdstC.resize((size_t)P, -1);   // P = end of this transform's logical block
        // ---- forward blocks
        int o; int64_t g;
        o = hereC(); g = gcur; gcur += (int64_t)NT * NI * 256; fwd_block(NT, NI, w_in, g);
        if (t == 0) { c.o_win = o; c.g_win = (int)g; }
        o = hereC(); g = gcur; gcur += NT * 16; bias_block(NT, hid_bias(lb0), hid_bias(lbc), g);
        if (t == 0) { c.o_b0 = o; c.g_b0 = (int)g; }
        o = hereC(); g = gcur; gcur += (int64_t)NT * NT * 256; fwd_block(NT, NT, w_hid(lWk[0]), g);
        if (t == 0) { c.o_w1 = o; c.g_w1 = (int)g; }
        o = hereC(); g = gcur; gcur += NT * 16; bias_block(NT, hid_bias(lbk[0]), nullptr, g);
        if (t == 0) { c.o_b1 = o; c.g_b1 = (int)g; }
        o = hereC(); g = gcur; gcur += (int64_t)NT * NT * 256; fwd_block(NT, NT, w_hid(lWk[1]), g);
        if (t == 0) { c.o_w2 = o; c.g_w2 = (int)g; }
        o = hereC(); g = gcur; gcur += NT * 16; bias_block(NT, hid_bias(lbk[1]), nullptr, g);
        if (t == 0) { c.o_b2 = o; c.g_b2 = (int)g; }
        o = hereC(); g = gcur; gcur += (int64_t)NT * 256; fwd_block(1, NT, w_head, g);
        if (t == 0) { c.o_wf = o; c.g_wf = (int)g; }
        o = hereC(); g = gcur; gcur += 16;
        bias_block(1, [&](int, int ro) -> int64_t { const int r_ = head_row(ro); return r_ >= 0 ? lbf + r_ : -1; }, nullptr, g);
        if (t == 0) { c.o_bf = o; c.g_bf = (int)g; }
        // ---- transposed blocks (data gradients)
        o = hereC(); tr_block(NT, 1, w_head);
        if (t == 0) c.o_wfT = o;
        o = hereC(); tr_block(NT, NT, w_hid(lWk[1]));
        if (t == 0) c.o_w2T = o;
        o = hereC(); tr_block(NT, NT, w_hid(lWk[0]));
        if (t == 0) c.o_w1T = o;
        o = hereC(); tr_block(NI, NT, w_in);
        if (t == 0) c.o_winT = o;
        if (t == 0) {
          c.g_stride = (int)((gcur + 63) / 64 * 64);
          // (the gdstC entries written above used g_stride = 0 for t = 0: gbase_t = 0 either way)
        }
      }
    }
  } else {
    Emitter ES{L.src16a, L.src16b, nullptr};  // NSF sampler image, fp32 part (sf_layout.h, SfNsfSamp)
    SfNsfSamp& sS = L.nsfS;
    sS.ok = d.hidden_bf16 ? 0 : 1;
    v.PT = K <= 11 ? 2 : 3;  // PT=1 (K<=5) is not instantiated: K<=11 shares the 2-tile layout
    v.KMAX = (v.PT * 16 + 1) / 3;
    const int dtr_max = (D + 1) / 2;
    v.JP = ceil_div(dtr_max, 2);
    const int NP = 3 * K - 1;
    for (int t = 0; t < T; ++t) {
      const int64_t tb = E.pad_to(64);
      if (t == 1) v.t_stride = (int)tb;
      std::vector<int> idn, tr;
      for (int i = 0; i < D; ++i) (((i % 2) == (t % 2)) ? tr : idn).push_back(i);
      const int d_id = (int)idn.size(), d_tr = (int)tr.size();
      const int in_dim = d_id + C;
      const int64_t lWin = P, lbin = lWin + (int64_t)H * in_dim;
      int64_t cur = lbin + H;
      int64_t lWg[SF_NBMAX], lbg[SF_NBMAX], lW1[SF_NBMAX], lb1[SF_NBMAX], lW2[SF_NBMAX], lb2[SF_NBMAX];
      for (int k = 0; k < NB; ++k) {
        lWg[k] = cur; lbg[k] = lWg[k] + (int64_t)H * C;
        lW1[k] = lbg[k] + H; lb1[k] = lW1[k] + (int64_t)H * H;
        lW2[k] = lb1[k] + H; lb2[k] = lW2[k] + (int64_t)H * H;
        cur = lb2[k] + H;
      }
      const int64_t lWout = cur, lbout = lWout + (int64_t)d_tr * NP * H;
      cur = lbout + (int64_t)d_tr * NP;
      int64_t lLo = -1, lUp = -1, lDi = -1, lBi = -1;
      if (D > 1) {
        const int nl = D * (D - 1) / 2;
        lLo = cur; lUp = lLo + nl; lDi = lUp + nl; lBi = lDi + D;
        cur = lBi + D;
      }
      P = cur;

      std::vector<int> urow(v.nGu * 8, -1);  // u tile row = phys slot = logical dim
      for (int j = 0; j < d_id; ++j) urow[idn[j]] = j;
      std::vector<int> crow(v.nGc * 8, -1);
      for (int i = 0; i < C; ++i) crow[i] = d_id + i;
      // spline head rows: [jp][pt][32]
      std::vector<int> orow(v.JP * v.PT * 32, -1);
      for (int jp = 0; jp < v.JP; ++jp)
        for (int pt = 0; pt < v.PT; ++pt)
          for (int rho = 0; rho < 32; ++rho) {
            int h = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
            int s = pt * 16 + r, kdim = 2 * jp + h, row = -1;
            if (kdim < d_tr) {
              if (s < v.KMAX) { if (s < K) row = kdim * NP + s; }
              else if (s < 2 * v.KMAX) { if (s - v.KMAX < K) row = kdim * NP + K + (s - v.KMAX); }
              else if (s < 3 * v.KMAX - 1) { if (s - 2 * v.KMAX < K - 1) row = kdim * NP + 2 * K + (s - 2 * v.KMAX); }
            }
            orow[(jp * v.PT + pt) * 32 + rho] = row;
          }
      int o;
      o = (int)(E.linear(HT, v.nGu, hrow_out, urow, lWin, in_dim, 1, nullptr) - tb);
      if (t == 0) v.o_winu = o;
      o = (int)(E.linear(HT, v.nGc, hrow_out, crow, lWin, in_dim, 1, nullptr) - tb);
      if (t == 0) v.o_winc = o;
      o = (int)(E.bias(HT, hrow_out, lbin, -1) - tb);
      if (t == 0) v.o_bin = o;
      for (int k = 0; k < NB; ++k) {
        o = (int)(E.linear(HT, v.nGc, hrow_out, crow_in, lWg[k], C, 1, nullptr) - tb);
        if (t == 0) v.o_wg[k] = o;
        o = (int)(E.bias(HT, hrow_out, lbg[k], -1) - tb);
        if (t == 0) v.o_bg[k] = o;
        o = (int)(E.linear(HT, v.nGh, hrow_out, hrow_in, lW1[k], H, 1, nullptr) - tb);
        if (t == 0) v.o_w1[k] = o;
        o = (int)(E.bias(HT, hrow_out, lb1[k], -1) - tb);
        if (t == 0) v.o_b1[k] = o;
        o = (int)(E.linear(HT, v.nGh, hrow_out, hrow_in, lW2[k], H, 1, nullptr) - tb);
        if (t == 0) v.o_w2[k] = o;
        o = (int)(E.bias(HT, hrow_out, lb2[k], -1) - tb);
        if (t == 0) v.o_b2[k] = o;
        if (v.hidden_bf16) {
          if (k == 0) {
            while (L.srcB.size() % 64) L.srcB.push_back(-1);
            if (t == 1) v.tB_stride = (int)L.srcB.size();
          }
          const int64_t tbB = (int64_t)t * (t >= 1 ? v.tB_stride : 0);
          const int64_t s1 = emit_bf16(lW1[k], nullptr);
          const int64_t s2 = emit_bf16(lW2[k], nullptr);
          if (t == 0) { v.oB_w1[k] = (int)(s1 - tbB); v.oB_w2[k] = (int)(s2 - tbB); }
        }
      }
      o = (int)(E.linear(v.JP * v.PT, v.nGh, orow, hrow_in, lWout, H, 1, nullptr) - tb);
      if (t == 0) v.o_wout = o;
      o = (int)(E.bias(v.JP * v.PT, orow, lbout, -1) - tb);
      if (t == 0) v.o_bout = o;
      // LU block: L[D*D] (strict lower), U[D*D] (strict upper), udiag[D], bias[D]
      {
        int64_t s = E.pad_to(4);
        if (t == 0) v.o_lu = (int)(s - tb);
        std::vector<int32_t> Lm(D * D, -1), Um(D * D, -1);
        if (D > 1) {
          int n = 0;
          for (int i = 0; i < D; ++i)
            for (int j = 0; j < i; ++j) Lm[i * D + j] = (int32_t)(lLo + n++);
          n = 0;
          for (int i = 0; i < D; ++i)
            for (int j = i + 1; j < D; ++j) Um[i * D + j] = (int32_t)(lUp + n++);
        }
        for (int i = 0; i < D * D; ++i) E.push(Lm[i], -1);
        for (int i = 0; i < D * D; ++i) E.push(Um[i], -1);
        for (int i = 0; i < D; ++i) E.push(D > 1 ? (int32_t)(lDi + i) : -1, -1);
        for (int i = 0; i < D; ++i) E.push(D > 1 ? (int32_t)(lBi + i) : -1, -1);
      }
      // ---- sampler image: the same blocks without W1 / W2, and those as split bf16
      if (sS.ok) {
        const int64_t tbS = ES.pad_to(64);
        if (t == 1) sS.t_stride = (int)tbS;
        int oS;
        oS = (int)(ES.linear(HT, v.nGu, hrow_out, urow, lWin, in_dim, 1, nullptr) - tbS);
        if (t == 0) sS.o_winu = oS;
        oS = (int)(ES.linear(HT, v.nGc, hrow_out, crow, lWin, in_dim, 1, nullptr) - tbS);
        if (t == 0) sS.o_winc = oS;
        oS = (int)(ES.bias(HT, hrow_out, lbin, -1) - tbS);
        if (t == 0) sS.o_bin = oS;
        for (int k = 0; k < NB; ++k) {
          oS = (int)(ES.linear(HT, v.nGc, hrow_out, crow_in, lWg[k], C, 1, nullptr) - tbS);
          if (t == 0) sS.o_wg[k] = oS;
          oS = (int)(ES.bias(HT, hrow_out, lbg[k], -1) - tbS);
          if (t == 0) sS.o_bg[k] = oS;
          oS = (int)(ES.bias(HT, hrow_out, lb1[k], -1) - tbS);
          if (t == 0) sS.o_b1[k] = oS;
          oS = (int)(ES.bias(HT, hrow_out, lb2[k], -1) - tbS);
          if (t == 0) sS.o_b2[k] = oS;
        }
        oS = (int)(ES.linear(v.JP * v.PT, v.nGh, orow, hrow_in, lWout, H, 1, nullptr) - tbS);
        if (t == 0) sS.o_wout = oS;
        oS = (int)(ES.bias(v.JP * v.PT, orow, lbout, -1) - tbS);
        if (t == 0) sS.o_bout = oS;
        {
          int64_t sl = ES.pad_to(4);
          if (t == 0) sS.o_lu = (int)(sl - tbS);
          std::vector<int32_t> Lm(D * D, -1), Um(D * D, -1);
          if (D > 1) {
            int n = 0;
            for (int i = 0; i < D; ++i)
              for (int j = 0; j < i; ++j) Lm[i * D + j] = (int32_t)(lLo + n++);
            n = 0;
            for (int i = 0; i < D; ++i)
              for (int j = i + 1; j < D; ++j) Um[i * D + j] = (int32_t)(lUp + n++);
          }
          for (int i = 0; i < D * D; ++i) ES.push(Lm[i], -1);
          for (int i = 0; i < D * D; ++i) ES.push(Um[i], -1);
          for (int i = 0; i < D; ++i) ES.push(D > 1 ? (int32_t)(lDi + i) : -1, -1);
          for (int i = 0; i < D; ++i) ES.push(D > 1 ? (int32_t)(lBi + i) : -1, -1);
        }
        // split-bf16 hidden blocks: [mt][ks][hi | lo][64 lanes][8]; element j of lane (c, h) = input row
        // 16 ks + 8 (j >> 2) + 4 h + (j & 3) (the order sf_bfrag hands the activations to v_mfma_f32_32x32x16_bf16)
        while (L.src16B.size() % 64) L.src16B.push_back(-1);
        const int64_t tbB = (int64_t)L.src16B.size();
        if (t == 1) sS.tB_stride = (int)tbB;
        auto emit_split = [&](int64_t base) -> int {
          const int64_t start = (int64_t)L.src16B.size() - (t >= 1 ? (int64_t)t * sS.tB_stride : 0);
          for (int mt = 0; mt < HT; ++mt)
            for (int ks = 0; ks < v.nKS; ++ks)
              for (int part = 0; part < 2; ++part)
                for (int l = 0; l < 64; ++l)
                  for (int j = 0; j < 8; ++j) {
                    const int o_ = hrow_full[mt * 32 + (l & 31)];
                    const int rho = 16 * ks + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
                    const int i_ = rho < 128 ? hrow_full[rho] : -1;
                    L.src16B.push_back((o_ >= 0 && i_ >= 0) ? (int32_t)((base + (int64_t)o_ * H + i_) | ((int64_t)part << 30)) : -1);
                  }
          return (int)start;
        };
        for (int k = 0; k < NB; ++k) {
          const int a1 = emit_split(lW1[k]), a2 = emit_split(lW2[k]);
          if (t == 0) { sS.oB_w1[k] = a1; sS.oB_w2[k] = a2; }
        }
      }
      // ---- transposed operands for the data-gradient pass ("o" = forward input, "i" = forward output)
      {
        const int64_t tbT = ET.pad_to(64);
        if (t == 1) v.tT_stride = (int)tbT;
        for (int jp = 0; jp < v.JP; ++jp) {
          std::vector<int> qrow(orow.begin() + jp * v.PT * 32, orow.begin() + (jp + 1) * v.PT * 32);
          o = (int)(ET.linear(HT, v.PT * 4, hrow_out, qrow, lWout, 1, H, nullptr) - tbT);
          if (t == 0 && jp == 0) v.oT_wout = o;
        }
        for (int k = 0; k < NB; ++k) {
          o = (int)(ET.linear(HT, v.nGh, hrow_out, hrow_in, lW2[k], 1, H, nullptr) - tbT);
          if (t == 0) v.oT_w2[k] = o;
          o = (int)(ET.linear(HT, v.nGh, hrow_out, hrow_in, lW1[k], 1, H, nullptr) - tbT);
          if (t == 0) v.oT_w1[k] = o;
        }
        std::vector<int> urow32(32, -1);
        for (int j = 0; j < d_id; ++j) urow32[idn[j]] = j;
        o = (int)(ET.linear(1, v.nGh, urow32, hrow_in, lWin, 1, in_dim, nullptr) - tbT);
        if (t == 0) v.oT_winu = o;
        const int CT = ceil_div(C, 32);
        std::vector<int> crow32(CT * 32, -1);
        for (int i = 0; i < C; ++i) crow32[i] = d_id + i;
        o = (int)(ET.linear(CT, v.nGh, crow32, hrow_in, lWin, 1, in_dim, nullptr) - tbT);
        if (t == 0) v.oT_wc = o;
        for (int k = 0; k < NB; ++k) {
          o = (int)(ET.linear(CT, v.nGh, iota_rows(C, CT * 32), hrow_in, lWg[k], 1, C, nullptr) - tbT);
          if (t == 0) v.oT_wg[k] = o;
        }
      }
      // ---- cooperative 16-row training image (sf_layout.h, SfNscDev; sf_nsfc.hip) -----------------------------
      if (NB == 2 && D >= 2 && D <= 8 && H <= 80 && 3 * K - 1 <= 32 && 8 + C <= 48) {
        SfNscDev& c = L.nsc;
        const int NT = ceil_div(H, 16), NI = ceil_div(8 + C, 16);
        const int OTQ = 3 * K - 1 <= 24 ? 6 : 8, KM = OTQ == 6 ? 8 : 11;
        if (t == 0) {
          c.ok = 1;
          c.NT = NT; c.NI = NI; c.OTQ = OTQ; c.KM = KM;
          c.kc_h = ceil_div(H - 16 * (NT - 1), 4);
          c.kc_in[0] = 4; c.kc_in[1] = c.kc_in[2] = 0;
          for (int it = 1; it < NI; ++it) c.kc_in[it] = ceil_div(std::min(16, C - 8 - 16 * (it - 1)), 4);
        }
        auto pushC = [&](int32_t a) { L.srcC1.push_back(a); L.srcC2.push_back(-1); };
        while (L.srcC1.size() % 256) pushC(-1);
        const int64_t tbC = (int64_t)L.srcC1.size();
        if (t == 1) c.t_stride = (int)tbC;
        auto hereC = [&]() { return (int)((int64_t)L.srcC1.size() - tbC); };
        int64_t gcur = 0;
        const int64_t gbase_t = (int64_t)t * c.g_stride;   // (g_stride is known from t = 0 on; 0 * anything for t = 0)
        if (t == 0) L.gdstC.assign((size_t)0, -1);
        if (L.gdstC.size() < (size_t)P) L.gdstC.resize((size_t)P, -1);
        // row maps (sf_layout.h)
        auto hid_unit = [&](int tile, int rho) -> int { const int u = tile * 16 + (rho >> 2) + 4 * (rho & 3); return u < H ? u : -1; };
        std::vector<int> id_col(8, -1);   // theta dimension -> column of W_in (identity dimensions only)
        for (int j = 0; j < d_id; ++j) id_col[idn[j]] = j;
        auto in_col = [&](int it, int rho, bool gate) -> int {   // column of W_in (gate: of W_g) behind an input-tile row
          const int g = rho >> 2, m = rho & 3;
          if (it == 0 && m < 2) { const int dd = 2 * g + m; return (!gate && dd < D) ? id_col[dd] : -1; }
          const int f = it == 0 ? (m - 2) * 4 + g : 8 + (it - 1) * 16 + 4 * m + g;
          if (f >= C) return -1;
          return gate ? f : d_id + f;
        };
        auto q_row = [&](int tile, int rho) -> int {   // row of W_out behind a spline-head tile row
          const int g = rho >> 2, sl = 4 * tile + (rho & 3);
          if (g >= d_tr) return -1;
          const int fam = sl / KM, kk = sl % KM;
          if (fam > 2 || kk >= (fam < 2 ? K : K - 1)) return -1;
          return g * NP + fam * K + kk;
        };
        using WFn = std::function<int64_t(int, int, int, int)>;   // (ot, ro, it, ri) -> logical index or -1
        auto fwd_block = [&](int OT, int IT, const WFn& fn, int64_t g_off) {
          for (int ot = 0; ot < OT; ++ot)
            for (int it = 0; it < IT; ++it)
              for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                  const int ro = l & 15, ri = 4 * (l >> 4) + r;
                  const int64_t idx = fn(ot, ro, it, ri);
                  pushC((int32_t)idx);
                  if (idx >= 0) L.gdstC[(size_t)idx] = (int32_t)(gbase_t + g_off + ((int64_t)ot * IT + it) * 256 + (ro & 3) * 64 + (ro >> 2) * 16 + ri);
                }
        };
        auto tr_block = [&](int ITr, int OTk, const WFn& fn) {   // rows = forward input tiles, K = forward output tiles
          for (int it = 0; it < ITr; ++it)
            for (int ot = 0; ot < OTk; ++ot)
              for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) pushC((int32_t)fn(ot, 4 * (l >> 4) + r, it, l & 15));
        };
        auto bias_block = [&](int OT, const std::function<int64_t(int, int)>& f1, int64_t g_off) {
          for (int ot = 0; ot < OT; ++ot)
            for (int ro = 0; ro < 16; ++ro) {
              const int64_t a1 = f1(ot, ro);
              pushC((int32_t)a1);
              if (a1 >= 0) L.gdstC[(size_t)a1] = (int32_t)(gbase_t + g_off + ot * 16 + ro);
            }
        };
        auto w_in = [&](int ot, int ro, int it, int ri) -> int64_t {
          const int oo = hid_unit(ot, ro), cc = in_col(it, ri, false);
          return (oo >= 0 && cc >= 0) ? lWin + (int64_t)oo * in_dim + cc : -1;
        };
        auto w_gate = [&](int64_t base) {
          return [&, base](int ot, int ro, int it, int ri) -> int64_t {
            const int oo = hid_unit(ot, ro), cc = in_col(it, ri, true);
            return (oo >= 0 && cc >= 0) ? base + (int64_t)oo * C + cc : -1;
          };
        };
        auto w_hid = [&](int64_t base) {
          return [&, base](int ot, int ro, int it, int ri) -> int64_t {
            const int oo = hid_unit(ot, ro), ii = hid_unit(it, ri);
            return (oo >= 0 && ii >= 0) ? base + (int64_t)oo * H + ii : -1;
          };
        };
        auto w_out = [&](int ot, int ro, int it, int ri) -> int64_t {
          const int oo = q_row(ot, ro), ii = hid_unit(it, ri);
          return (oo >= 0 && ii >= 0) ? lWout + (int64_t)oo * H + ii : -1;
        };
        auto hid_bias = [&](int64_t base) {
          return [&, base](int ot, int ro) -> int64_t { const int oo = hid_unit(ot, ro); return oo >= 0 ? base + oo : -1; };
        };
        int oC; int64_t g;
#define SF_NSC_FWD(field_o, field_g, OT_, IT_, fn)                                   \
  oC = hereC(); g = gcur; gcur += (int64_t)(OT_) * (IT_) * 256; fwd_block(OT_, IT_, fn, g); \
  if (t == 0) { c.field_o = oC; c.field_g = (int)g; }
#define SF_NSC_BIAS(field_o, field_g, OT_, fn)                                       \
  oC = hereC(); g = gcur; gcur += (int64_t)(OT_) * 16; bias_block(OT_, fn, g);       \
  if (t == 0) { c.field_o = oC; c.field_g = (int)g; }
        SF_NSC_FWD(o_win, g_win, NT, NI, w_in)
        SF_NSC_BIAS(o_bin, g_bin, NT, hid_bias(lbin))
        for (int k = 0; k < 2; ++k) {
          SF_NSC_FWD(o_wg[k], g_wg[k], NT, NI, w_gate(lWg[k]))
          SF_NSC_BIAS(o_bg[k], g_bg[k], NT, hid_bias(lbg[k]))
          SF_NSC_FWD(o_w1[k], g_w1[k], NT, NT, w_hid(lW1[k]))
          SF_NSC_BIAS(o_b1[k], g_b1[k], NT, hid_bias(lb1[k]))
          SF_NSC_FWD(o_w2[k], g_w2[k], NT, NT, w_hid(lW2[k]))
          SF_NSC_BIAS(o_b2[k], g_b2[k], NT, hid_bias(lb2[k]))
        }
        SF_NSC_FWD(o_wout, g_wout, OTQ, NT, w_out)
        SF_NSC_BIAS(o_bout, g_bout, OTQ, [&](int ot, int ro) -> int64_t { const int oo = q_row(ot, ro); return oo >= 0 ? lbout + oo : -1; })
#undef SF_NSC_FWD
#undef SF_NSC_BIAS
        // LU block: L[8][8], U[8][8], udiag[8], bias[8]; gradient: two 16 x 16 blocks + 16 row sums
        {
          oC = hereC(); g = gcur; gcur += 2 * 256 + 16;
          if (t == 0) { c.o_lu = oC; c.g_lu = (int)g; }
          auto gpos = [&](int blk, int ro, int ri) { return (int32_t)(gbase_t + g + blk * 256 + (ro & 3) * 64 + (ro >> 2) * 16 + ri); };
          std::vector<int32_t> Lm(64, -1), Um(64, -1);
          int n = 0;
          for (int i = 0; i < D; ++i)
            for (int j = 0; j < i; ++j) { Lm[i * 8 + j] = (int32_t)(lLo + n); L.gdstC[(size_t)(lLo + n)] = gpos(0, i, j); ++n; }
          n = 0;
          for (int i = 0; i < D; ++i)
            for (int j = i + 1; j < D; ++j) { Um[i * 8 + j] = (int32_t)(lUp + n); L.gdstC[(size_t)(lUp + n)] = gpos(1, i, j); ++n; }
          for (int i = 0; i < 64; ++i) pushC(Lm[i]);
          for (int i = 0; i < 64; ++i) pushC(Um[i]);
          for (int i = 0; i < 8; ++i) { pushC(i < D ? (int32_t)(lDi + i) : -1); if (i < D) L.gdstC[(size_t)(lDi + i)] = (int32_t)(gbase_t + g + 512 + 8 + i); }
          for (int i = 0; i < 8; ++i) { pushC(i < D ? (int32_t)(lBi + i) : -1); if (i < D) L.gdstC[(size_t)(lBi + i)] = (int32_t)(gbase_t + g + 512 + i); }
          while ((L.srcC1.size() - (size_t)tbC) % 16) pushC(-1);
        }
        // transposed blocks (data gradients)
        oC = hereC(); tr_block(NT, OTQ, w_out);
        if (t == 0) c.o_woutT = oC;
        for (int k = 0; k < 2; ++k) {
          oC = hereC(); tr_block(NT, NT, w_hid(lW2[k]));
          if (t == 0) c.o_w2T[k] = oC;
          oC = hereC(); tr_block(NT, NT, w_hid(lW1[k]));
          if (t == 0) c.o_w1T[k] = oC;
        }
        oC = hereC(); tr_block(1, NT, w_in);   // only input tile 0 holds theta dimensions (no context gradient on this path)
        if (t == 0) c.o_winT = oC;
        if (t == 0) c.g_stride = (int)((gcur + 63) / 64 * 64);
      }
    }
  }
  E.pad_to(64);
  ET.pad_to(64);
  while (L.srcB.size() % 64) L.srcB.push_back(-1);
  while (L.src16a.size() % 1024) { L.src16a.push_back(-1); L.src16b.push_back(-1); }
  if (T == 1) {
    v.t_stride = (int)E.cur; v.tT_stride = (int)ET.cur; v.tB_stride = (int)L.srcB.size();
    v.t16_stride = (int)L.src16a.size();
  }
  if (d.kind == SF_NSF && L.nsfS.ok) {
    while (L.src16a.size() % 64) { L.src16a.push_back(-1); L.src16b.push_back(-1); }
    while (L.src16B.size() % 64) L.src16B.push_back(-1);
    if (T == 1) { L.nsfS.t_stride = (int)L.src16a.size(); L.nsfS.tB_stride = (int)L.src16B.size(); }
    // staging plan: whole items (input layer | block k: fp32 pieces + split W1 / W2 | head) packed into the LDS budget next to the
    // sampler's control block; at most three parts, no item larger than the budget
    if (L.nsfS.tB_stride & 7) L.nsfS.ok = 0;
    if (L.nsfS.ok) {
      SfNsfSamp& sS = L.nsfS;
      const size_t budget = (size_t)152 * 1024;
      std::vector<int> cf{0}, cb{0};   // item boundaries: fp32 floats, bf16 elements
      for (int k = 0; k < NB; ++k) { cf.push_back(sS.o_wg[k]); cb.push_back(sS.oB_w1[k]); }
      cf.push_back(sS.o_wout); cb.push_back(sS.tB_stride);
      cf.push_back(sS.t_stride); cb.push_back(sS.tB_stride);
      sS.n_parts = 0; sS.part_off[0] = 0; sS.partB_off[0] = 0;
      std::vector<int> item_part(cf.size() - 1, 0);
      int sf0 = 0, sb0 = 0;
      bool okp = true;
      for (size_t i = 0; i + 1 < cf.size() && okp; ++i) {
        const size_t item = (size_t)(cf[i + 1] - cf[i]) * 4 + (size_t)(cb[i + 1] - cb[i]) * 2;
        if (item > budget) { okp = false; break; }
        if ((size_t)(cf[i + 1] - sf0) * 4 + (size_t)(cb[i + 1] - sb0) * 2 > budget) {   // close the current part before this item
          if (sS.n_parts >= 3) { okp = false; break; }
          ++sS.n_parts;
          sS.part_off[sS.n_parts] = cf[i]; sS.partB_off[sS.n_parts] = cb[i];
          sf0 = cf[i]; sb0 = cb[i];
        }
        item_part[i] = sS.n_parts;
      }
      if (okp) {
        ++sS.n_parts;
        sS.part_off[sS.n_parts] = cf.back(); sS.partB_off[sS.n_parts] = cb.back();
        sS.part_floats_max = 0; sS.part_bytes_max = 0;
        for (int p = 0; p < sS.n_parts; ++p) {
          const int pf = sS.part_off[p + 1] - sS.part_off[p], pb = sS.partB_off[p + 1] - sS.partB_off[p];
          sS.part_floats_max = std::max(sS.part_floats_max, pf);
          sS.part_bytes_max = std::max(sS.part_bytes_max, pf * 4 + pb * 2);
          if ((sS.partB_off[p] & 7) || (sS.part_off[p] & 3)) okp = false;   // 16-byte pieces of the direct-to-LDS copies
        }
        for (int k = 0; k < NB; ++k) sS.blk_part[k] = item_part[1 + k];
        sS.head_part = item_part[1 + NB];
      }
      if (!okp) sS.ok = 0;
    }
    if (!L.nsfS.ok) { L.src16a.clear(); L.src16b.clear(); L.src16B.clear(); }
  }
  L.n_packed16 = (int64_t)L.src16a.size();
  if (d.kind == SF_MAF) while (L.src16B.size() % 2048) L.src16B.push_back(-1);
  if (T == 1) v.t16B_stride = (int)(L.src16B.size() / 2);
  L.n_packed16B = (int64_t)L.src16B.size();
  if (!v.m16_ok && !L.nsfS.ok) { L.src16B.clear(); L.n_packed16B = 0; }
  if (v.m16_ok && (size_t)v.t16_stride * sizeof(float) > 152 * 1024) v.m16_ok = 0;
  L.n_packedB = (int64_t)L.srcB.size();
  // ---- LDS staging plan (budget: 152 KiB of the 160 KiB LDS; the rest holds the persistent sampler's control block) ---------------------------------
  {
    const int budget = 152 * 1024 / 4;
    std::vector<int> cuts;  // block boundaries (float offsets) in execution order, first = 0, last = end
    if (d.kind == SF_MAF) {
      cuts = {0, v.t_stride};
    } else {
      cuts.push_back(0);
      for (int k = 0; k < NB; ++k) cuts.push_back(v.o_wg[k]);  // block k starts at its gate weights
      cuts.push_back(v.o_wout);
      cuts.push_back(v.t_stride);  // spline head + the small LU block (LU is used from LDS only when n_parts == 1)
    }
    v.n_parts = 0;
    v.part_max = 0;
    bool ok = true;
    int start = 0;
    std::vector<int> item_part(cuts.size() - 1, 0);
    v.part_off[0] = 0;
    for (size_t i = 0; i + 1 < cuts.size(); ++i) {
      const int item = cuts[i + 1] - cuts[i];
      if (item > budget) { ok = false; break; }
      if (cuts[i + 1] - start > budget) {  // close the current part before this item
        if (v.n_parts >= 3) { ok = false; break; }
        v.part_off[++v.n_parts] = cuts[i];
        start = cuts[i];
      }
      item_part[i] = v.n_parts;
    }
    if (ok) {
      v.part_off[++v.n_parts] = cuts.back();
      for (int p = 0; p < v.n_parts; ++p) v.part_max = std::max(v.part_max, v.part_off[p + 1] - v.part_off[p]);
      if (d.kind == SF_NSF) {
        for (int k = 0; k < NB; ++k) v.blk_part[k] = item_part[1 + k];
        v.head_part = item_part[1 + NB];
      }
    } else {
      v.n_parts = 0;
    }
    for (int p = 0; p < 5; ++p) v.partB_off[p] = p == 0 ? 0 : v.tB_stride;   // (single-part bf16 images: everything with part 0)
    v.part_bytes_max = v.part_max * 4 + (v.hidden_bf16 ? v.tB_stride * 2 : 0);
    if (v.hidden_bf16) {
      const int need = v.t_stride + (v.tB_stride + 1) / 2;  // floats: fp32 image + bf16 image of one transform
      if (v.n_parts != 1 || need > budget)
        return fail("hidden_bf16: one transform's fp32 + bf16 operand images must fit the 152 KiB LDS budget");
    }
  }
  if (L.trc.ok || L.nsc.ok) {
    while (L.srcC1.size() % 256) { L.srcC1.push_back(-1); L.srcC2.push_back(-1); }
    if (T == 1) { L.trc.t_stride = (int)L.srcC1.size(); L.nsc.t_stride = (int)L.srcC1.size(); }
    L.n_imgC = (int64_t)L.srcC1.size();
    L.n_gradC = (int64_t)T * (L.trc.ok ? L.trc.g_stride : L.nsc.g_stride);
    L.gdstC.resize((size_t)P, -1);
  }
  L.n_packed = E.cur;
  L.n_packedT = ET.cur;
  L.n_params = P;
  L.gdst.assign((size_t)P, -1);
  for (int64_t i = 0; i < E.cur; ++i) {
    if (L.src1[i] >= 0 && L.gdst[L.src1[i]] < 0) L.gdst[L.src1[i]] = gidx[i];  // first image copy owns the gradient
    if (L.src2[i] >= 0 && L.gdst[L.src2[i]] < 0) L.gdst[L.src2[i]] = gidx[i];
  }
  return true;
}

// =============================================================================================
// embedding MLP
// =============================================================================================
bool sf_build_mlp_layout(const sf_mlp_desc& d, SfMlpLayout& L) {
  auto fail = [&](const std::string& m) { L.error = m; return false; };
  if (d.n_in < 1 || d.n_in > 512) return fail("n_in must be in 1..512");
  if (d.n_layers < 1 || d.n_layers > SF_MLP_LMAX) return fail("n_layers must be in 1..4");
  if (d.act < 0 || d.act > 2) return fail("unknown activation");
  int wmax = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    if (d.widths[l] < 1 || d.widths[l] > 128) return fail("layer widths must be in 1..128");
    wmax = std::max(wmax, d.widths[l]);
  }
  SfMlpDev& v = L.dev;
  std::memset(&v, 0, sizeof(v));
  v.n_in = d.n_in; v.L = d.n_layers; v.act = d.act; v.n_out = d.widths[d.n_layers - 1];
  v.HT = ceil_div(wmax, 32);
  L.cst.assign(2 * ((d.n_in + 3) / 4) * 4, 1.f);
  for (int i = 0; i < d.n_in; ++i) {
    L.cst[i] = d.x_mean ? d.x_mean[i] : 0.f;
    L.cst[((d.n_in + 3) / 4) * 4 + i] = d.x_std ? d.x_std[i] : 1.f;
  }
  std::vector<int32_t> gidx;
  Emitter E{L.src1, L.src2, &gidx};
  Emitter ET{L.srcT1, L.srcT2, nullptr};
  int64_t P = 0;
  int in_w = d.n_in;
  for (int l = 0; l < d.n_layers; ++l) {
    const int out_w = d.widths[l];
    v.width[l] = out_w;
    v.nG[l] = ceil_div(in_w, 8);
    v.nGo[l] = ceil_div(out_w, 8);
    const int64_t lW = P, lb = lW + (int64_t)out_w * in_w;
    P = lb + out_w;
    E.pad_to(64);
    v.o_w[l] = (int)E.linear(v.HT, v.nG[l], iota_rows(out_w, v.HT * 32), iota_rows(in_w, v.nG[l] * 8), lW, in_w, 1, nullptr);
    v.o_b[l] = (int)E.bias(v.HT, iota_rows(out_w, v.HT * 32), lb, -1);
    if (l > 0) {  // delta_in = W^T delta_out : rows = previous layer's units, K = this layer's outputs
      ET.pad_to(64);
      v.oT_w[l] = (int)ET.linear(v.HT, v.nGo[l], iota_rows(in_w, v.HT * 32), iota_rows(out_w, v.nGo[l] * 8), lW, 1,
                                 in_w, nullptr);
    }
    in_w = out_w;
  }
  E.pad_to(64);
  ET.pad_to(64);
  L.n_packed = E.cur;
  L.n_packedT = ET.cur;
  L.n_params = P;
  L.gdst.assign((size_t)P, -1);
  for (int64_t i = 0; i < E.cur; ++i)
    if (L.src1[i] >= 0) L.gdst[L.src1[i]] = gidx[i];
  return true;
}
