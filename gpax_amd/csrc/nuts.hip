// nuts.hip — one NUTS transition of the exact-GP posterior over (log) hyper-parameters on the HOST side of the library
// (include/gpx.h gpx_nuts_transition): leapfrogs, recursive doubling, multinomial / biased progressive sampling and the
// generalised U-turn test in C++ around the device fit step (gpx_fit_batch with B = 1: ONE kernel launch up to N = 128,
// csrc/fit_small.hip).
//
// Role on the path: numpyro.infer.NUTS as ExactGP.fit drives it (gpax/models/gp.py:207-218) at the sizes every reference
// notebook runs (examples/gpax_simpleGP.ipynb:232: N = 25).  There a leapfrog is 22 us of device work under ~80 us of
// host time, half of it the Python tree building of gpax_amd/infer/nuts.py (profiles/r05: nuts_overhead) — this file is
// that loop, statement for statement, for the default model: every site LogNormal-distributed (gp.py:222-247), no mean
// function.  Adaptation (dual averaging, mass-matrix windows, the step-size search), custom priors and mean functions
// stay in Python on the existing path.
//
// Same chain as the Python loop: the uniforms come from the caller's generator — NumPy's PCG64 (128-bit LCG, XSL-RR
// output, doubles = 53 high bits), advanced here and handed back — in the very order gpax_amd/infer/nuts.py draws
// them; the momentum is drawn by the caller (Generator.standard_normal is a ziggurat over the same stream).  The floating-
// point operations follow the Python statements.  Driven with this file's potential (gpx_nuts_potential) and its dot
// products summed in order, the Python loop produces THIS loop's chain bit for bit (tests/test_gpu_nuts_native.py); in
// production its potential goes through NumPy, whose exp / log (SIMD) and dot (BLAS) round the last bit differently from
// libm in a few per cent of the calls: the same trees, positions equal to rounding per transition — and, a chain being a
// chaotic map, to ~1e-5 after a few hundred transitions.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

// every statement below is meant as written: no fused multiply-add where the source has a product and a sum (the Python
// loop this restates rounds each operation)
#pragma clang fp contract(off)

namespace {

constexpr double MAX_DELTA_ENERGY = 1000.0;
constexpr int MAXD = GPX_MAX_DIM + 4; // u: lengthscales (+ period) + scale + noise

// ---- numpy.random.PCG64 -----------------------------------------------------------------------------------------------
struct Pcg64 {
  unsigned __int128 state, inc;
  uint64_t next64() {
    const unsigned __int128 mult = ((unsigned __int128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    state = state * mult + inc;
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const unsigned rot = (unsigned)(state >> 122);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64u - rot) & 63u));
  }
  double uniform() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
};

struct Vec {
  double v[MAXD];
};

struct Model {
  gpx_ctx* ctx;
  int kind, dim, ne; // ne = entries of the device's lengthscale vector (d, + 1: period)
  const int* idx_ell; // u index of every device lengthscale entry
  int idx_scale, idx_noise;
  const double *loc, *scale, *cst; // per element of u: LogNormal(loc, scale), cst = log(scale) + log(2 pi) / 2
  double jitter;
  const double* yres;
  const double* inv_mass;
  int n_eval = 0;
  int rc = 0;
};

// numpy's pairwise summation for n < 128 (what ndarray.sum() does on these vectors)
double np_sum(const double* a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r[8];
  for (int k = 0; k < 8; ++k) r[k] = a[k];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int k = 0; k < 8; ++k) r[k] += a[i + k];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}

// potential U(u) = -(lml(theta(u)) + log prior + log |d theta / d u|) and its gradient: ExactGP._log_joint +
// ExactGP._chain_rule (the all-LogNormal plan) + the sign flip of fit()'s `potential` (gpax_amd/models/gp.py)
void potential(Model& m, const Vec& u, double& U, Vec& g) {
  const int dim = m.dim;
  double x[MAXD], lx[MAXD], term[MAXD];
  for (int i = 0; i < dim; ++i) {
    x[i] = std::exp(u.v[i]);
    lx[i] = std::log(x[i]);
  }
  double ell[GPX_MAX_DIM + 1];
  for (int c = 0; c < m.ne; ++c) ell[c] = x[m.idx_ell[c]];
  const double sc = x[m.idx_scale], nz = x[m.idx_noise];
  double lml = 0.0, grad[GPX_MAX_DIM + 3];
  int info = 0;
  m.n_eval += 1;
  const int rc = gpx_fit_batch(m.ctx, m.kind, 1, ell, &sc, &nz, m.jitter, m.yres, 1, &lml, &info, grad, nullptr);
  if (rc != 0) {
    m.rc = rc;
    info = 1;
  }
  for (int i = 0; i < dim; ++i) g.v[i] = 0.0;
  if (info != 0 || !std::isfinite(lml)) {
    U = INFINITY;
    return;
  }
  for (int i = 0; i < dim; ++i) {
    const double z = (lx[i] - m.loc[i]) / m.scale[i];
    term[i] = -0.5 * z * z - m.cst[i] - lx[i];
  }
  double val = lml + np_sum(term, dim);
  val += np_sum(u.v, dim);
  if (!std::isfinite(val)) {
    U = INFINITY;
    return;
  }
  double gx[MAXD];
  for (int c = 0; c < m.ne; ++c) gx[m.idx_ell[c]] = grad[c];
  gx[m.idx_scale] = grad[m.ne];
  gx[m.idx_noise] = grad[m.ne + 1];
  for (int i = 0; i < dim; ++i) {
    const double gu = (gx[i] + (-(lx[i] - m.loc[i]) / (m.scale[i] * m.scale[i]) - 1.0) / x[i]) * x[i] + 1.0;
    g.v[i] = -gu;
  }
  U = -val;
}

double dot_w(const Model& m, const Vec& a, const Vec& b) { // a @ (inv_mass * b), summed as (a * (inv_mass * b)).sum()
  double t[MAXD];
  for (int i = 0; i < m.dim; ++i) t[i] = a.v[i] * (m.inv_mass[i] * b.v[i]);
  return np_sum(t, m.dim);
}

double energy(const Model& m, double U, const Vec& p) {
  if (!std::isfinite(U)) return INFINITY;
  return U + 0.5 * dot_w(m, p, p);
}

void leapfrog(Model& m, Vec& u, Vec& p, double& U, Vec& g, double eps) {
  const double he = 0.5 * eps;
  for (int i = 0; i < m.dim; ++i) p.v[i] = p.v[i] - he * g.v[i];
  for (int i = 0; i < m.dim; ++i) u.v[i] = u.v[i] + (eps * m.inv_mass[i]) * p.v[i];
  potential(m, u, U, g);
  for (int i = 0; i < m.dim; ++i) p.v[i] = p.v[i] - he * g.v[i];
}

bool uturn(const Model& m, const Vec& rho, const Vec& pl, const Vec& pr) {
  Vec r;
  for (int i = 0; i < m.dim; ++i) r.v[i] = rho.v[i] - 0.5 * (pl.v[i] + pr.v[i]);
  return (dot_w(m, r, pl) <= 0.0) || (dot_w(m, r, pr) <= 0.0);
}

double logaddexp(double a, double b) { // numpy.logaddexp
  if (a == b) return a + 0.69314718055994530942;
  const double t = a - b;
  if (t > 0) return a + std::log1p(std::exp(-t));
  if (t <= 0) return b + std::log1p(std::exp(t));
  return t; // NaN
}

struct Tree {
  Vec ul, pl, gl, ur, pr, gr;
  Vec prop_u, prop_g;
  double prop_U;
  double logw;
  Vec rho;
  bool turning, diverging;
  double sum_accept;
  int n;
};

// gpax_amd/infer/nuts.py _build_tree
void build_tree(Model& m, const Vec& u, const Vec& p, const Vec& g, int direction, int depth, double eps, double H0,
                Pcg64& rng, Tree& out) {
  if (depth == 0) {
    Vec u1 = u, p1 = p, g1 = g;
    double U1 = 0.0;
    leapfrog(m, u1, p1, U1, g1, direction * eps);
    const double H1 = energy(m, U1, p1);
    double dH = H1 - H0;
    if (std::isnan(dH)) dH = INFINITY;
    out.ul = out.ur = out.prop_u = u1;
    out.pl = out.pr = out.rho = p1;
    out.gl = out.gr = out.prop_g = g1;
    out.prop_U = U1;
    out.logw = -dH;
    out.turning = false;
    out.diverging = dH > MAX_DELTA_ENERGY;
    out.sum_accept = std::fmin(1.0, std::exp(std::fmin(0.0, -dH)));
    out.n = 1;
    return;
  }
  build_tree(m, u, p, g, direction, depth - 1, eps, H0, rng, out);
  if (out.turning || out.diverging) return;
  Tree* b = new Tree;
  if (direction == 1) build_tree(m, out.ur, out.pr, out.gr, direction, depth - 1, eps, H0, rng, *b);
  else build_tree(m, out.ul, out.pl, out.gl, direction, depth - 1, eps, H0, rng, *b);
  const double logw = logaddexp(out.logw, b->logw);
  if (!(b->turning || b->diverging)) {
    if (std::log(rng.uniform()) < b->logw - logw) { // multinomial within the new subtree pair
      out.prop_u = b->prop_u;
      out.prop_g = b->prop_g;
      out.prop_U = b->prop_U;
    }
  }
  for (int i = 0; i < m.dim; ++i) out.rho.v[i] = out.rho.v[i] + b->rho.v[i];
  if (direction == 1) {
    out.ur = b->ur;
    out.pr = b->pr;
    out.gr = b->gr;
  } else {
    out.ul = b->ul;
    out.pl = b->pl;
    out.gl = b->gl;
  }
  out.turning = b->turning || uturn(m, out.rho, out.pl, out.pr);
  out.diverging = b->diverging;
  out.logw = logw;
  out.sum_accept += b->sum_accept;
  out.n += b->n;
  delete b;
}

} // namespace

extern "C" {

int gpx_debug_pcg64_doubles(uint64_t* state4, int n, double* out) {
  if (!state4 || n < 0 || (n > 0 && !out)) return -1;
  Pcg64 r;
  r.state = ((unsigned __int128)state4[0] << 64) | state4[1];
  r.inc = ((unsigned __int128)state4[2] << 64) | state4[3];
  for (int i = 0; i < n; ++i) out[i] = r.uniform();
  state4[0] = (uint64_t)(r.state >> 64);
  state4[1] = (uint64_t)r.state;
  return 0;
}

static int check_model_args(gpx_ctx* ctx, int kind, int dim, int ne, const int* idx_ell, int idx_scale, int idx_noise) {
  const int ne_want = ctx->d + (kind == GPX_KERNEL_PERIODIC ? 1 : 0);
  if (ne != ne_want || dim != ne + 2 || dim > MAXD) return gpx::bad_arg(ctx, "NUTS model: dim must be d (+ 1: period) + 2");
  std::vector<char> seen((size_t)dim, 0);
  auto mark = [&](int i) -> bool {
    if (i < 0 || i >= dim || seen[(size_t)i]) return false;
    seen[(size_t)i] = 1;
    return true;
  };
  bool ok = mark(idx_scale) && mark(idx_noise);
  for (int c = 0; ok && c < ne; ++c) ok = mark(idx_ell[c]);
  if (!ok) return gpx::bad_arg(ctx, "NUTS model: the index maps must be a permutation of 0 .. dim - 1");
  return 0;
}

int gpx_nuts_potential(gpx_ctx* ctx, int kind, int dim, int ne, const int* idx_ell, int idx_scale, int idx_noise,
                       const double* prior_loc, const double* prior_scale, const double* prior_const, double jitter,
                       const double* yres, const double* u, double* U, double* g) {
  if (!ctx || ctx->device < 0) return -1;
  if (ctx->N < 1) return gpx::bad_arg(ctx, "gpx_set_train must be called first");
  if (!idx_ell || !prior_loc || !prior_scale || !prior_const || !yres || !u || !U || !g) return gpx::bad_arg(ctx, "null pointer");
  const int rc = check_model_args(ctx, kind, dim, ne, idx_ell, idx_scale, idx_noise);
  if (rc != 0) return rc;
  Model m{ctx, kind, dim, ne, idx_ell, idx_scale, idx_noise, prior_loc, prior_scale, prior_const, jitter, yres, nullptr};
  Vec uu{}, gg{};
  for (int i = 0; i < dim; ++i) uu.v[i] = u[i];
  potential(m, uu, *U, gg);
  for (int i = 0; i < dim; ++i) g[i] = gg.v[i];
  return m.rc;
}

int gpx_nuts_transition(gpx_ctx* ctx, int kind, int dim, int ne, const int* idx_ell, int idx_scale, int idx_noise,
                        const double* prior_loc, const double* prior_scale, const double* prior_const, double jitter,
                        const double* yres, double* u, double* U, double* g, const double* p0, double eps,
                        const double* inv_mass, int max_tree_depth, uint64_t* pcg_state4, double* accept, int* n_leapfrog,
                        int* diverging) {
  if (!ctx || ctx->device < 0) return -1;
  if (ctx->N < 1) return gpx::bad_arg(ctx, "gpx_set_train must be called first");
  if (!idx_ell || !prior_loc || !prior_scale || !prior_const || !yres || !u || !U || !g || !p0 || !inv_mass || !pcg_state4 ||
      !accept || !n_leapfrog || !diverging)
    return gpx::bad_arg(ctx, "null pointer");
  {
    const int rc = check_model_args(ctx, kind, dim, ne, idx_ell, idx_scale, idx_noise);
    if (rc != 0) return rc;
  }
  if (max_tree_depth < 1 || max_tree_depth > 20) return gpx::bad_arg(ctx, "max_tree_depth");
  Model m{ctx, kind, dim, ne, idx_ell, idx_scale, idx_noise, prior_loc, prior_scale, prior_const, jitter, yres, inv_mass};
  Pcg64 rng;
  rng.state = ((unsigned __int128)pcg_state4[0] << 64) | pcg_state4[1];
  rng.inc = ((unsigned __int128)pcg_state4[2] << 64) | pcg_state4[3];

  // gpax_amd/infer/nuts.py nuts_transition
  Vec u0{}, g0{}, pm{};
  for (int i = 0; i < dim; ++i) {
    u0.v[i] = u[i];
    g0.v[i] = g[i];
    pm.v[i] = p0[i];
  }
  const double H0 = energy(m, *U, pm);
  Vec ul = u0, ur = u0, pl = pm, pr = pm, gl = g0, gr = g0, rho = pm;
  Vec prop_u = u0, prop_g = g0;
  double prop_U = *U, logw = 0.0, sum_accept = 0.0;
  int n_leap = 0;
  bool div = false;
  Tree* t = new Tree;
  for (int depth = 0; depth < max_tree_depth; ++depth) {
    const int direction = (rng.uniform() < 0.5) ? 1 : -1;
    if (direction == 1) {
      build_tree(m, ur, pr, gr, 1, depth, eps, H0, rng, *t);
      ur = t->ur;
      pr = t->pr;
      gr = t->gr;
    } else {
      build_tree(m, ul, pl, gl, -1, depth, eps, H0, rng, *t);
      ul = t->ul;
      pl = t->pl;
      gl = t->gl;
    }
    sum_accept += t->sum_accept;
    n_leap += t->n;
    if (t->diverging) {
      div = true;
      break;
    }
    if (t->turning) break;
    if (std::log(rng.uniform()) < t->logw - logw) { // biased progressive sampling
      prop_u = t->prop_u;
      prop_g = t->prop_g;
      prop_U = t->prop_U;
    }
    logw = logaddexp(logw, t->logw);
    for (int i = 0; i < dim; ++i) rho.v[i] = rho.v[i] + t->rho.v[i];
    if (uturn(m, rho, pl, pr)) break;
  }
  delete t;
  if (m.rc != 0) return m.rc; // a device error inside a leapfrog (gpx_last_error has the message)
  for (int i = 0; i < dim; ++i) {
    u[i] = prop_u.v[i];
    g[i] = prop_g.v[i];
  }
  *U = prop_U;
  *accept = sum_accept / (n_leap > 1 ? n_leap : 1);
  *n_leapfrog = n_leap;
  *diverging = div ? 1 : 0;
  pcg_state4[0] = (uint64_t)(rng.state >> 64);
  pcg_state4[1] = (uint64_t)rng.state;
  return 0;
}

} // extern "C"
