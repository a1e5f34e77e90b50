/* sanitize_main.c -- ASAN + UBSAN exercise of the oracle's C restatement (oracle/c/pdec_oracle.c).
 * TEST INFRASTRUCTURE ONLY: built by `make -C oracle/c sanitize` into oracle/_build/oracle_sanitize and run by
 * tests/test_oracle.py on the CPU (GPU-side sanitizers are not available on this pool).  It drives every exported
 * entry point at ragged sizes -- mixed-radix transform lengths (the reference's 192 / 240 grids and the bench's 256),
 * odd batch sizes, sensor windows that wrap, 2- and 3-layer nets, a minibatch of 3 and one of 77 columns -- with the
 * arrays allocated at their EXACT sizes, so any out-of-bounds index, use-after-free, signed overflow or misaligned
 * access of the restatement is reported; exit code 0 = clean. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int N, S, A, window; double max_value, agent_power, action_punish, delta_action_punish;
                 const double* G; const double* Ga; const int* a2s; } env_tabs;
void* ks_plan_create(int N, double Lx, double dt, int K, double mu);
void ks_plan_destroy(void* p);
void ks_step(void* plan, const double* y, const double* p, double* yout);
void ks_env_step_batch(void* plan, const env_tabs* T, int B, double* y, const double* action, const double* action_prev,
                       double* state_out, double* reward_out, int* done);
void* agent_create(int La, const int* dimsA, const int* actsA, int Lc, const int* dimsC, const int* actsC);
void agent_set(void* g, int which, const double* flat);
void agent_get(void* g, int which, double* flat);
int agent_nparams(void* g, int which);
void agent_act(void* g, const double* state, const double* noise, int cols, double act_noise, double lim, double* actions);
void agent_ddpg_update(void* g, const double* s, const double* a, const double* r, const double* t, const double* snext, int Bu,
                       double gamma, double rho, int quirk, double eta_a, double eta_c, double* actor_loss, double* critic_loss);
void oracle_set_threads(int n);
int oracle_num_threads(void);

static unsigned long long rs = 88172645463325252ull;
static double urand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0 * 2 - 1; }
static double* rvec(size_t n, double scale) { double* v = malloc(8 * n); for (size_t i = 0; i < n; ++i) v[i] = scale * urand(); return v; }
static int finite_all(const double* v, size_t n) { for (size_t i = 0; i < n; ++i) if (!isfinite(v[i])) return 0; return 1; }

static int run_env(int N, int stride, int window, int B) {
  const int S = (N + stride - 1) / stride, A = S;
  void* plan = ks_plan_create(N, N * (200.0 / 240.0), 0.1, 30, 0.01);
  if (!plan) { fprintf(stderr, "plan N=%d refused\n", N); return 1; }
  double* G = calloc((size_t)S * N, 8); double* Ga = calloc((size_t)A * N, 8); int* a2s = malloc(4 * A);
  for (int s = 0; s < S; ++s) { a2s[s] = s;
    for (int d = -3; d <= 3; ++d) { const int c = ((s * stride + d) % N + N) % N; G[(size_t)s * N + c] = 0.2 / (1 + d * d); Ga[(size_t)s * N + c] = 1.0 / (1 + d * d); } }
  env_tabs T = {N, S, A, window, 30.0, 7.5, 0.002, 0.002, G, Ga, a2s};
  double* y = rvec((size_t)B * N, 0.5); double* a = rvec((size_t)B * A, 1.0); double* ap = rvec((size_t)B * A, 1.0);
  double* st = malloc(8 * (size_t)B * A * window); double* rw = malloc(8 * (size_t)B * A); int* done = malloc(4 * B);
  double* p = rvec(N, 0.1); double* yo = malloc(8 * N);
  ks_step(plan, y, p, yo);
  int bad = !finite_all(yo, N);
  for (int it = 0; it < 2; ++it) ks_env_step_batch(plan, &T, B, y, a, ap, st, rw, done);
  bad |= !finite_all(y, (size_t)B * N) || !finite_all(st, (size_t)B * A * window) || !finite_all(rw, (size_t)B * A);
  free(G); free(Ga); free(a2s); free(y); free(a); free(ap); free(st); free(rw); free(done); free(p); free(yo);
  ks_plan_destroy(plan);
  return bad;
}

static int run_agent(int ns, int h, int H, int three, int Bu) {
  int dA3[4] = {ns, h, h, 1}, aA3[3] = {1, 1, 2}, dC3[4] = {ns + 1, H, H, 1}, aC3[3] = {1, 1, 0};
  int dA2[3] = {ns, h, 1}, aA2[2] = {1, 2}, dC2[3] = {ns + 1, H, 1}, aC2[2] = {1, 0};
  void* g = three ? agent_create(3, dA3, aA3, 3, dC3, aC3) : agent_create(2, dA2, aA2, 2, dC2, aC2);
  for (int w = 0; w < 4; ++w) { const int n = agent_nparams(g, w); double* f = rvec(n, 0.3); agent_set(g, w, f); agent_get(g, w, f); free(f); }
  double* s = rvec((size_t)Bu * ns, 1); double* sn = rvec((size_t)Bu * ns, 1); double* noise = rvec(Bu, 1);
  double* act = malloc(8 * Bu); double* r = rvec(Bu, 1); double* t = calloc(Bu, 8);
  t[Bu - 1] = 1.0;
  agent_act(g, s, noise, Bu, 0.3, 1.0, act);
  double la = 0, lc = 0;
  for (int quirk = 0; quirk < 2; ++quirk) agent_ddpg_update(g, s, act, r, t, sn, Bu, 0.99, 0.995, quirk, 5e-4, 1e-3, &la, &lc);
  const int n = agent_nparams(g, 1); double* f = malloc(8 * n); agent_get(g, 1, f);
  int bad = !finite_all(f, n) || !isfinite(la) || !isfinite(lc) || !finite_all(act, Bu);
  free(f); free(s); free(sn); free(noise); free(act); free(r); free(t);
  /* the restatement has no agent destructor (the Python wrapper keeps agents for the life of the process); the leak
   * checker is told so in the Makefile (detect_leaks=0) rather than papering over it here */
  return bad;
}

int main(void) {
  int bad = 0;
  for (int threads = 1; threads <= 3; threads += 2) {
    oracle_set_threads(threads);
    bad |= run_env(192, 24, 1, 1);       /* KS22 geometry, single trajectory */
    bad |= run_env(240, 3, 1, 5);        /* KS200: 80 sensors, odd batch */
    bad |= run_env(256, 4, 3, 7);        /* bench C2 geometry, window 3 wraps around */
    bad |= run_env(60, 7, 5, 2);         /* ragged: radices 4*3*5, sensors not dividing N, window 5 */
    bad |= run_agent(1, 6, 140, 0, 3);   /* the shipped 2-layer nets, minibatch of 3 */
    bad |= run_agent(3, 16, 140, 1, 77); /* C2's 3-layer nets, ragged column count */
  }
  if (oracle_num_threads() < 1) bad = 1;
  printf(bad ? "sanitize: FAILED\n" : "sanitize: OK\n");
  return bad;
}
