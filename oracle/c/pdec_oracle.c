/* pdec_oracle.c -- plain-C fp64 restatement of the reference algorithm for one env step +
 * one DDPG update.  TEST INFRASTRUCTURE ONLY: the checker for tests/ and the timed
 * `cpu_baseline` ("port") of bench.py; never on the product path.  PINNED: ks_step is
 * checked against the reference's golden trajectories (tests/test_oracle_c.py, <= 1e-12);
 * the NN part is checked against oracle/nn.py (itself pinned by FD + torch autograd).
 *
 * Restates (paths relative to janstenner/DistributedConvRL-PDE-Control):
 *   KS do_step (CNAB2)         scripts/KS/setup/KSSetup.jl:115-160
 *   prepare_action             scripts/KS/setup/KSSetup.jl:231-245
 *   reward_function            scripts/KS/setup/KSSetup.jl:162-178
 *   featurize                  scripts/KS/setup/KSSetup.jl:190-229
 *   policy act                 src/PDEagent.jl:183-207
 *   DDPG update, ADAM, Polyak  src/PDEagent.jl:363-418
 * The reference re-evaluates fft(env.p) and fft(mu cos ...) inside every sub-step
 * (KSSetup.jl:155); they are loop invariants and are hoisted here (2K+3 FFTs per step),
 * which only makes this baseline faster than the reference's own code.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_max_threads(void) { return 1; }
static int omp_get_thread_num(void) { return 0; }
#endif

typedef double complex cpx;

/* ---------------- mixed-radix (2,3,4,5) Stockham FFT, unnormalised both ways ---------- */
typedef struct { int N, ns, radix[16]; cpx* tw; } fft_plan;

static int plan_init(fft_plan* pl, int N) {
  pl->N = N; pl->ns = 0;
  int n = N; const int cand[4] = {4, 2, 3, 5};
  for (int c = 0; c < 4; ++c) { int r = cand[c];
    while (n % r == 0 && !(r == 2 && n % 4 == 0)) { pl->radix[pl->ns++] = r; n /= r; } }
  if (n != 1) return -1;
  pl->tw = (cpx*)malloc(sizeof(cpx) * N);
  for (int k = 0; k < N; ++k) pl->tw[k] = cexp(-2.0 * M_PI * I * k / N);
  return 0;
}

/* x is overwritten with the transform; y is scratch of N elements */
static void fft_run(const fft_plan* pl, cpx* x, cpx* y, int sign) {
  const int N = pl->N; int n = N, s = 1; cpx *X = x, *Y = y;
  for (int st = 0; st < pl->ns; ++st) {
    const int r = pl->radix[st], m = n / r;
    for (int t = 0; t < N / r; ++t) {
      const int p = t / s, q = t % s; cpx a[5], b[5];
      for (int j = 0; j < r; ++j) a[j] = X[q + s * (p + m * j)];
      for (int k = 0; k < r; ++k) { cpx acc = 0;
        for (int j = 0; j < r; ++j) { cpx w = pl->tw[((long)j * k % r) * (N / r)]; if (sign > 0) w = conj(w); acc += a[j] * w; }
        b[k] = acc; }
      for (int k = 0; k < r; ++k) { cpx w = pl->tw[p * s * k]; if (sign > 0) w = conj(w); Y[q + s * (r * p + k)] = b[k] * w; }
    }
    cpx* T = X; X = Y; Y = T; n /= r; s *= r;
  }
  if (X != x) memcpy(x, X, sizeof(cpx) * N);
}

/* ---------------- KS CNAB2 control step, KSSetup.jl:130-160 --------------------------- */
typedef struct {
  int N, K; double Lx, dt, mu; fft_plan fft;
  double *Ainv, *Bc, *alpha; cpx* G; cpx* Dhat;
} ks_plan;

void* ks_plan_create(int N, double Lx, double dt, int K, double mu) {
  ks_plan* P = (ks_plan*)calloc(1, sizeof(ks_plan));
  P->N = N; P->K = K; P->Lx = Lx; P->dt = dt; P->mu = mu;
  if (plan_init(&P->fft, N)) { free(P); return NULL; }
  P->Ainv = malloc(8 * N); P->Bc = malloc(8 * N); P->alpha = malloc(8 * N);
  P->G = malloc(sizeof(cpx) * N); P->Dhat = malloc(sizeof(cpx) * N);
  const double h = dt / K, dt2 = h / 2, dx = Lx / N;
  cpx* scr = malloc(sizeof(cpx) * N);
  for (int k = 0; k < N; ++k) {
    double kx = k < N / 2 ? k : (k == N / 2 ? 0 : k - N);     /* :115 Nyquist slot holds 0 */
    double al = 2 * M_PI * kx / Lx, L = al * al - al * al * al * al;   /* :116-118 */
    P->alpha[k] = al; P->G[k] = -0.5 * I * al;                 /* :119 */
    P->Ainv[k] = 1.0 / (1.0 - dt2 * L); P->Bc[k] = 1.0 + dt2 * L;  /* :134-135 */
    P->Dhat[k] = mu * cos(2 + M_PI + dx * (k + 1) / (Lx / 2));  /* :155 */
  }
  fft_run(&P->fft, P->Dhat, scr, -1);
  for (int k = 0; k < N; ++k) P->Dhat[k] *= h;
  free(scr);
  return P;
}
void ks_plan_destroy(void* p) { ks_plan* P = p; if (!P) return; free(P->Ainv); free(P->Bc); free(P->alpha); free(P->G); free(P->Dhat); free(P->fft.tw); free(P); }

/* work: 5*N complex */
static void ks_step_w(const ks_plan* P, const double* y, const double* p, double* yout, cpx* work) {
  const int N = P->N; const double h = P->dt / P->K, dt2 = h / 2, dt32 = 3 * h / 2;
  cpx *u = work, *Nn = work + N, *Nn1 = work + 2 * N, *Ph = work + 3 * N, *scr = work + 4 * N;
  for (int n = 0; n < N; ++n) { u[n] = y[n]; Nn[n] = y[n] * y[n]; Ph[n] = p[n]; }
  fft_run(&P->fft, Nn, scr, -1);
  for (int k = 0; k < N; ++k) Nn[k] *= P->G[k];               /* :140 */
  fft_run(&P->fft, u, scr, -1);                                /* :142 */
  fft_run(&P->fft, Ph, scr, -1);
  for (int it = 0; it < P->K; ++it) {
    memcpy(Nn1, Nn, sizeof(cpx) * N);                          /* :145 */
    memcpy(Nn, u, sizeof(cpx) * N);                            /* :146 */
    fft_run(&P->fft, Nn, scr, +1);                             /* :148 */
    for (int n = 0; n < N; ++n) { cpx w = Nn[n] / N; Nn[n] = w * w; }   /* :149 (w kept complex) */
    fft_run(&P->fft, Nn, scr, -1);                             /* :150 */
    for (int k = 0; k < N; ++k) {
      Nn[k] *= P->G[k];                                        /* :152 */
      u[k] = P->Ainv[k] * (P->Bc[k] * u[k] + dt32 * Nn[k] - dt2 * Nn1[k] + h * Ph[k]) + P->Dhat[k];  /* :155 */
    }
  }
  fft_run(&P->fft, u, scr, +1);                                /* :158 */
  for (int n = 0; n < N; ++n) yout[n] = creal(u[n]) / N;       /* :159 */
}

void ks_step(void* plan, const double* y, const double* p, double* yout) {
  ks_plan* P = plan; cpx* w = malloc(sizeof(cpx) * 5 * P->N);
  ks_step_w(P, y, p, yout, w); free(w);
}

/* ---------------- full env step for a batch (OpenMP over trajectories) ----------------- */
typedef struct {
  int N, S, A, window; double max_value, agent_power, action_punish, delta_action_punish;
  const double* G;   /* [S][N] sensor kernels */
  const double* Ga;  /* [A][N] actuator kernels */
  const int* a2s;    /* [A] 0-based */
} env_tabs;

/* y [B][N] in/out, action/action_prev [B][A], state_out [B][A][ns], reward_out [B][A], done [B] */
void ks_env_step_batch(void* plan, const env_tabs* T, int B, double* y, const double* action, const double* action_prev,
                       double* state_out, double* reward_out, int* done) {
  ks_plan* P = plan; const int N = T->N, S = T->S, A = T->A, ns = T->window, w = T->window / 2;
#pragma omp parallel
  {
    cpx* work = malloc(sizeof(cpx) * 5 * N);
    double* p = malloc(8 * N); double* yn = malloc(8 * N); double* sens = malloc(8 * S);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      const double* a = action + (size_t)b * A; const double* ap = action_prev + (size_t)b * A;
      memset(p, 0, 8 * N);
      for (int i = 0; i < A; ++i) { const double c = T->agent_power * a[i]; const double* g = T->Ga + (size_t)i * N;
        for (int n = 0; n < N; ++n) p[n] += c * g[n]; }        /* prepare_action, :241 */
      ks_step_w(P, y + (size_t)b * N, p, yn, work);
      double mx = 0;
      for (int n = 0; n < N; ++n) { y[(size_t)b * N + n] = yn[n]; if (fabs(yn[n]) > mx) mx = fabs(yn[n]); }
      done[b] = mx > T->max_value;                              /* PDEenv.jl:227 */
      for (int s = 0; s < S; ++s) { double acc = 0; const double* g = T->G + (size_t)s * N;
        for (int n = 0; n < N; ++n) acc += yn[n] * g[n]; sens[s] = acc; }
      for (int i = 0; i < A; ++i) {
        const double d = 6.0 * sens[T->a2s[i]], da = a[i] - ap[i];           /* :163-169 */
        reward_out[(size_t)b * A + i] = -pow(fabs(d), 1.3) / (T->max_value * 3) - T->action_punish * a[i] * a[i] -
                                        T->delta_action_punish * da * da;   /* :178 */
        for (int r = 0; r < ns; ++r) { int s = (T->a2s[i] - (r - w)) % S; if (s < 0) s += S;   /* :205-207 */
          state_out[((size_t)b * A + i) * ns + r] = sens[s] / T->max_value; }                    /* :201 */
      }
    }
    free(work); free(p); free(yn); free(sens);
  }
}

/* ---------------- MLP (<= 3 layers), DDPG update -------------------------------------- */
#define MAXL 4
typedef struct { int L, dims[MAXL + 1], acts[MAXL]; double* W[MAXL]; double* b[MAXL]; } mlp;   /* W[l] row-major [out][in] */
static double actf(double z, int k) { return k == 1 ? (z > 0 ? z : 0) : (k == 2 ? tanh(z) : z); }
static double dactf(double a, int k) { return k == 1 ? (a > 0 ? 1 : 0) : (k == 2 ? 1 - a * a : 1); }

/* one column forward; h[l] = activations (h[0] = x) */
static void mlp_fwd1(const mlp* M, double* const* h) {
  for (int l = 0; l < M->L; ++l) { const int in = M->dims[l], out = M->dims[l + 1];
    for (int o = 0; o < out; ++o) { double acc = 0; const double* w = M->W[l] + (size_t)o * in; const double* x = h[l];
      /* the dot product in SIMD lanes (a reassociated sum: ~1e-16 relative, far inside the 1e-12 the twin is held to against the
       * NumPy oracle); without the pragma the scalar reduction runs at a quarter of the AVX2 rate */
#pragma omp simd reduction(+:acc)
      for (int i = 0; i < in; ++i) acc += w[i] * x[i];
      h[l + 1][o] = actf(acc + M->b[l][o], M->acts[l]); } }
}
/* one column backward; d[L] = dL/da_L on entry; accumulates gW/gb if non-NULL; leaves dL/dx in d[0] */
static void mlp_bwd1(const mlp* M, double* const* h, double* const* d, double* const* gW, double* const* gb) {
  for (int l = M->L - 1; l >= 0; --l) { const int in = M->dims[l], out = M->dims[l + 1];
    for (int i = 0; i < in; ++i) d[l][i] = 0;
    for (int o = 0; o < out; ++o) { const double dz = d[l + 1][o] * dactf(h[l + 1][o], M->acts[l]);
      const double* w = M->W[l] + (size_t)o * in;
      if (gW) { double* g = gW[l] + (size_t)o * in; for (int i = 0; i < in; ++i) g[i] += dz * h[l][i]; gb[l][o] += dz; }
      for (int i = 0; i < in; ++i) d[l][i] += w[i] * dz; } }
}

typedef struct { mlp A, C, At, Ct; double *mA, *vA, *mC, *vC; double bpA[2], bpC[2]; int nA, nC; } agent;

static int mlp_alloc(mlp* M, int L, const int* dims, const int* acts) {
  M->L = L; int n = 0;
  for (int l = 0; l <= L; ++l) M->dims[l] = dims[l];
  for (int l = 0; l < L; ++l) { M->acts[l] = acts[l]; M->W[l] = calloc((size_t)dims[l] * dims[l + 1], 8); M->b[l] = calloc(dims[l + 1], 8);
    n += dims[l] * dims[l + 1] + dims[l + 1]; }
  return n;
}
/* params flat: W1 row-major [out][in], b1, ... */
static void mlp_set(mlp* M, const double* flat) { size_t o = 0; for (int l = 0; l < M->L; ++l) { size_t nw = (size_t)M->dims[l] * M->dims[l + 1];
    memcpy(M->W[l], flat + o, 8 * nw); o += nw; memcpy(M->b[l], flat + o, 8 * M->dims[l + 1]); o += M->dims[l + 1]; } }
static void mlp_get(const mlp* M, double* flat) { size_t o = 0; for (int l = 0; l < M->L; ++l) { size_t nw = (size_t)M->dims[l] * M->dims[l + 1];
    memcpy(flat + o, M->W[l], 8 * nw); o += nw; memcpy(flat + o, M->b[l], 8 * M->dims[l + 1]); o += M->dims[l + 1]; } }

void* agent_create(int La, const int* dimsA, const int* actsA, int Lc, const int* dimsC, const int* actsC) {
  agent* G = calloc(1, sizeof(agent));
  G->nA = mlp_alloc(&G->A, La, dimsA, actsA); mlp_alloc(&G->At, La, dimsA, actsA);
  G->nC = mlp_alloc(&G->C, Lc, dimsC, actsC); mlp_alloc(&G->Ct, Lc, dimsC, actsC);
  G->mA = calloc(G->nA, 8); G->vA = calloc(G->nA, 8); G->mC = calloc(G->nC, 8); G->vC = calloc(G->nC, 8);
  G->bpA[0] = G->bpC[0] = 0.9; G->bpA[1] = G->bpC[1] = 0.999;
  return G;
}
void agent_set(void* g, int which, const double* flat) { agent* G = g; mlp_set(which == 0 ? &G->A : which == 1 ? &G->C : which == 2 ? &G->At : &G->Ct, flat); }
void agent_get(void* g, int which, double* flat) { agent* G = g; mlp_get(which == 0 ? &G->A : which == 1 ? &G->C : which == 2 ? &G->At : &G->Ct, flat); }
int agent_nparams(void* g, int which) { agent* G = g; return (which == 0 || which == 2) ? G->nA : G->nC; }

/* actions[cols][na] = clamp(A(state) + noise*act_noise) ; state [cols][ns]   (PDEagent.jl:183-207) */
void agent_act(void* g, const double* state, const double* noise, int cols, double act_noise, double lim, double* actions) {
  agent* G = g; const mlp* M = &G->A; const int ns = M->dims[0], na = M->dims[M->L];
#pragma omp parallel
  { double* h[MAXL + 1]; for (int l = 0; l <= M->L; ++l) h[l] = malloc(8 * M->dims[l]);
#pragma omp for schedule(static)
    for (int c = 0; c < cols; ++c) { memcpy(h[0], state + (size_t)c * ns, 8 * ns); mlp_fwd1(M, h);
      for (int f = 0; f < na; ++f) { double v = h[M->L][f] + (noise ? noise[(size_t)c * na + f] * act_noise : 0);
        actions[(size_t)c * na + f] = v < -lim ? -lim : (v > lim ? lim : v); } }
    for (int l = 0; l <= M->L; ++l) free(h[l]); }
}

static void adam(double* const* W, double* const* b, const mlp* M, double* flatg, double* m, double* v, double* bp, double eta) {
  const double b1 = 0.9, b2 = 0.999, eps = 1e-8; size_t o = 0;
  for (int l = 0; l < M->L; ++l) for (int part = 0; part < 2; ++part) {
    size_t n = part == 0 ? (size_t)M->dims[l] * M->dims[l + 1] : (size_t)M->dims[l + 1]; double* p = part == 0 ? W[l] : b[l];
    for (size_t i = 0; i < n; ++i, ++o) { const double g = flatg[o];
      m[o] = b1 * m[o] + (1 - b1) * g; v[o] = b2 * v[o] + (1 - b2) * g * g;
      p[i] -= m[o] / (1 - bp[0]) / (sqrt(v[o] / (1 - bp[1])) + eps) * eta; } }
  bp[0] *= b1; bp[1] *= b2;
}

/* gradient pass over columns with thread-private accumulators.  mode 0: critic loss, 1: actor loss */
static void grads_pass(agent* G, int mode, const double* s, const double* a, const double* dq_or_null, int Bu, double* flatg, double* qout) {
  const mlp *A = &G->A, *C = &G->C; const mlp* M = mode == 0 ? C : A; const int n = mode == 0 ? G->nC : G->nA;
  const int ns = A->dims[0], na = A->dims[A->L];
  /* per-thread partial gradients, combined in THREAD ORDER (deterministic for a given thread count; the static schedule
   * gives thread k the k-th contiguous block of columns), the combination itself parallel over the parameters */
  const int T = omp_get_max_threads();
  double* part = calloc((size_t)T * n, 8);
#pragma omp parallel
  {
    double* loc = part + (size_t)omp_get_thread_num() * n; double *gW[MAXL], *gb[MAXL]; size_t o = 0;
    for (int l = 0; l < M->L; ++l) { gW[l] = loc + o; o += (size_t)M->dims[l] * M->dims[l + 1]; gb[l] = loc + o; o += M->dims[l + 1]; }
    double *hc[MAXL + 1], *dc[MAXL + 1], *ha[MAXL + 1], *da[MAXL + 1];
    for (int l = 0; l <= C->L; ++l) { hc[l] = malloc(8 * C->dims[l]); dc[l] = malloc(8 * C->dims[l]); }
    for (int l = 0; l <= A->L; ++l) { ha[l] = malloc(8 * A->dims[l]); da[l] = malloc(8 * A->dims[l]); }
#pragma omp for schedule(static)
    for (int c = 0; c < Bu; ++c) {
      if (mode == 0) {
        memcpy(hc[0], s + (size_t)c * ns, 8 * ns); memcpy(hc[0] + ns, a + (size_t)c * na, 8 * na);
        mlp_fwd1(C, hc); dc[C->L][0] = dq_or_null[c]; mlp_bwd1(C, hc, dc, gW, gb);
      } else {
        memcpy(ha[0], s + (size_t)c * ns, 8 * ns); mlp_fwd1(A, ha);
        memcpy(hc[0], s + (size_t)c * ns, 8 * ns); memcpy(hc[0] + ns, ha[A->L], 8 * na);
        mlp_fwd1(C, hc); if (qout) qout[c] = hc[C->L][0];
        dc[C->L][0] = -1.0 / Bu; mlp_bwd1(C, hc, dc, NULL, NULL);
        memcpy(da[A->L], dc[0] + ns, 8 * na); mlp_bwd1(A, ha, da, gW, gb);
      }
    }
    /* (implicit barrier of the loop above: every partial is complete) */
#pragma omp for schedule(static)
    for (int i = 0; i < n; ++i) { double acc = 0; for (int k = 0; k < T; ++k) acc += part[(size_t)k * n + i]; flatg[i] = acc; }
    for (int l = 0; l <= C->L; ++l) { free(hc[l]); free(dc[l]); }
    for (int l = 0; l <= A->L; ++l) { free(ha[l]); free(da[l]); }
  }
  free(part);
}

/* one DDPG update, PDEagent.jl:363-418.  s,snext [Bu][ns]; a [Bu][na]; r,t [Bu] */
void agent_ddpg_update(void* g, const double* s, const double* a, const double* r, const double* t, const double* snext, int Bu,
                       double gamma, double rho, int quirk, double eta_a, double eta_c, double* actor_loss, double* critic_loss) {
  agent* G = g; const int ns = G->A.dims[0], na = G->A.dims[G->A.L];
  double* q = malloc(8 * Bu); double* tgt = malloc(8 * Bu); double* dq = malloc(8 * Bu);
#pragma omp parallel
  { double *hc[MAXL + 1], *ha[MAXL + 1];
    for (int l = 0; l <= G->C.L; ++l) hc[l] = malloc(8 * G->C.dims[l]);
    for (int l = 0; l <= G->A.L; ++l) ha[l] = malloc(8 * G->A.dims[l]);
#pragma omp for schedule(static)
    for (int c = 0; c < Bu; ++c) {
      memcpy(ha[0], snext + (size_t)c * ns, 8 * ns); mlp_fwd1(&G->At, ha);                         /* :385 */
      memcpy(hc[0], snext + (size_t)c * ns, 8 * ns); memcpy(hc[0] + ns, ha[G->A.L], 8 * na); mlp_fwd1(&G->Ct, hc);   /* :386 */
      tgt[c] = gamma * (1 - t[c]) * hc[G->C.L][0];
      memcpy(hc[0], s + (size_t)c * ns, 8 * ns); memcpy(hc[0] + ns, a + (size_t)c * na, 8 * na); mlp_fwd1(&G->C, hc);
      q[c] = hc[G->C.L][0];
    }
    for (int l = 0; l <= G->C.L; ++l) free(hc[l]);
    for (int l = 0; l <= G->A.L; ++l) free(ha[l]); }
  double rm = 0, cm = 0, c2 = 0, r2 = 0, diag = 0;
  for (int c = 0; c < Bu; ++c) { const double cc = tgt[c] - q[c]; rm += r[c]; cm += cc; c2 += cc * cc; r2 += r[c] * r[c];
    diag += (r[c] + cc) * (r[c] + cc); }
  rm /= Bu; cm /= Bu; c2 /= Bu; r2 /= Bu; diag /= Bu;
  *critic_loss = quirk ? c2 + 2 * cm * rm + r2 : diag;                                               /* :393 */
  for (int c = 0; c < Bu; ++c) dq[c] = -(2.0 / Bu) * ((quirk ? rm : r[c]) + tgt[c] - q[c]);
  double* gc = malloc(8 * (size_t)G->nC); double* ga = malloc(8 * (size_t)G->nA);
  grads_pass(G, 0, s, a, dq, Bu, gc, NULL);
  adam(G->C.W, G->C.b, &G->C, gc, G->mC, G->vC, G->bpC, eta_c);                                      /* :400 */
  grads_pass(G, 1, s, NULL, NULL, Bu, ga, q);
  double al = 0; for (int c = 0; c < Bu; ++c) al += q[c]; *actor_loss = -al / Bu;                    /* :403 */
  adam(G->A.W, G->A.b, &G->A, ga, G->mA, G->vA, G->bpA, eta_a);                                      /* :412 */
  for (int l = 0; l < G->A.L; ++l) { size_t nw = (size_t)G->A.dims[l] * G->A.dims[l + 1];           /* :415-417 */
    for (size_t i = 0; i < nw; ++i) G->At.W[l][i] = rho * G->At.W[l][i] + (1 - rho) * G->A.W[l][i];
    for (int i = 0; i < G->A.dims[l + 1]; ++i) G->At.b[l][i] = rho * G->At.b[l][i] + (1 - rho) * G->A.b[l][i]; }
  for (int l = 0; l < G->C.L; ++l) { size_t nw = (size_t)G->C.dims[l] * G->C.dims[l + 1];
    for (size_t i = 0; i < nw; ++i) G->Ct.W[l][i] = rho * G->Ct.W[l][i] + (1 - rho) * G->C.W[l][i];
    for (int i = 0; i < G->C.dims[l + 1]; ++i) G->Ct.b[l][i] = rho * G->Ct.b[l][i] + (1 - rho) * G->C.b[l][i]; }
  free(q); free(tgt); free(dq); free(gc); free(ga);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void oracle_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
