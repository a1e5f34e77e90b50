/* pdeconv.h -- C ABI of libpdeconv.so: MI355X (gfx950) hot path of
 * janstenner/DistributedConvRL-PDE-Control, hand-written HIP behind plain C.
 *
 * The reference has NO FFI (pure Julia); the seam this library replaces is the set of
 * Julia callables that `PDEenv` / `CustomDDPGPolicy` invoke.  Each entry
 * point below cites the reference callable (file:line, relative to the reference repo) it
 * stands in for.  The Julia `ccall` bindings a maintainer would add are in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns int: 0 = OK, <0 = error (PDEC_E_*); text via pdec_last_error()
 *    (thread-local).  No exceptions cross the boundary.  Blow-up of the PDE is NOT an error:
 *    it sets the per-trajectory `done` flag (src/PDEenv.jl:226-237).
 *  - handles are opaque uint64_t created/destroyed in pairs; a handle is not re-entrant,
 *    distinct handles may be used from distinct threads.
 *  - array arguments of compute calls are DEVICE pointers (pdec_malloc, or any HIP
 *    allocation such as a torch tensor's data_ptr); calls are asynchronous on the handle's
 *    stream (pdec_set_stream; default = the null stream).  The `_host` wrappers take HOST
 *    pointers, stage through plan-owned buffers and return after completion -- these are
 *    what a Julia `do_step(env)` closure binds.
 *  - arrays are dense, batch-major `[B][...Julia column-major...]`: a Julia matrix
 *    `state[ns, A]` of trajectory b starts at `state + b*ns*A` and element (r,a) is at
 *    `a*ns + r`.  dtype (PDEC_F32 / PDEC_F64) is fixed at plan creation; setup-time tables
 *    (sensor kernels, weights in set/get) are passed as documented per call.
 */
#ifndef PDECONV_H
#define PDECONV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t pdec_handle;

enum { PDEC_F32 = 0, PDEC_F64 = 1 };

enum {
  PDEC_OK = 0,
  PDEC_E_INVALID = -1,   /* bad argument / unsupported size */
  PDEC_E_HIP = -2,       /* HIP runtime error */
  PDEC_E_HANDLE = -3,    /* unknown / wrong-kind handle */
  PDEC_E_NOGPU = -4,     /* no usable gfx950 device */
  PDEC_E_COMM = -5       /* RCCL error */
};

/* PDE right-hand side / integrator kinds */
enum {
  PDEC_PDE_KS_CNAB2 = 0,     /* scripts/KS/setup/KSSetup.jl:130-160 (what the reference runs) */
  PDEC_PDE_KSEG_RK4 = 1,     /* scripts/Keller-Segel/setup/KellerSegelSetup.jl:213-239 with fixed RK4 */
  PDEC_PDE_KS_RK4_FD = 2,    /* north-star variant: RK4 + periodic 5-point FD (KSSetup.jl:55-59 table) */
  PDEC_PDE_FLUID_RK4 = 3,    /* src/fluid_rk4.jl:122-190 + scripts/Fluid/setup/FluidSetup.jl:163-172 */
  PDEC_PDE_KSEG2D_RK4 = 4    /* BASELINE.json configs[3]: the Keller-Segel rules (KellerSegelSetup.jl:63-66,213-239)
                                along both axes of a 2-D grid; no reference counterpart */
};

enum { PDEC_ACT_IDENTITY = 0, PDEC_ACT_RELU = 1, PDEC_ACT_TANH = 2 };

/* ---------------------------------------------------------------- runtime ------------ */
int pdec_init(int device_ordinal);            /* selects the device, checks gfx950 */
int pdec_shutdown(void);                      /* destroys every live handle */
const char* pdec_last_error(void);
int pdec_version(void);
int pdec_device_count(int* n);

int pdec_malloc(void** dptr, size_t bytes);
int pdec_free(void* dptr);
int pdec_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int pdec_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int pdec_memset(void* dptr, int value, size_t bytes);
int pdec_set_stream(pdec_handle h, void* hip_stream);   /* hipStream_t; NULL = null stream */
/* A non-blocking stream at an explicit priority LEVEL (-1 high, 0 normal, +1 low; clamped to the device's range), made together
 * with its hardware queue.  Why a library call: on this GPU a process's hardware queues are spread over FOUR compute pipes in
 * the order they are made, queue i on pipe i mod 4, and two BUSY queues on one pipe take turns instead of running side by
 * side (tools/c4_stream_matrix.sh, 21 creation orders: config C4 runs at 75 k env-steps/s when the env, update and two
 * part-batch streams sit on four pipes and at 40 - 46 k when two of them share one; HIP also lets at most four queues exist
 * per level and maps further streams of that level onto those).  So the streams that work at the same time -- env stream,
 * update stream, the part streams of pdec_env_set_part_streams -- are made BACK TO BACK by one caller, at most four of them,
 * and kept for the life of the process (queues made after others were destroyed take over the freed slots in an order of
 * their own: a fresh set per pipeline, the old one released, was measured worse than one set used again and again).  The
 * first call also makes the null stream's queue if nothing has yet, so that it cannot land between the caller's later.  A
 * framework's pooled streams give no such control (torch makes each pool stream at its first use).  hipGraph launches make
 * queues of their own for parallel branches; with all four pipes taken by busy streams they share one (C2: graph replay 190 -
 * 260 instead of 125 us per step -- eager issue, 110 us, is what the pipeline uses).  The reference has no counterpart
 * (single stream). */
int pdec_stream_create(void** hip_stream, int level);
int pdec_stream_destroy(void* hip_stream);
int pdec_sync(pdec_handle h);                           /* hipStreamSynchronize of h's stream */
int pdec_destroy(pdec_handle h);                        /* any handle kind */

/* Per-kernel timing with HIP events on the handle's stream (for bench.py's roofline):
 * when enabled every launch of the handle's kernels is bracketed by events. */
int pdec_prof_enable(pdec_handle h, int on);
int pdec_prof_reset(pdec_handle h);
/* name: kernel label, e.g. "ks_env_step", "ddpg_critic"; returns mean ms and launch count */
int pdec_prof_get(pdec_handle h, const char* name, double* mean_ms, int* count);

/* ---------------------------------------------------------------- environment -------- */
/* Configuration of one batched PDE environment = the globals of a reference setup file
 * (scripts/KS/setup/KSSetup.jl:20-77, scripts/Keller-Segel/setup/KellerSegelSetup.jl:26-84)
 * plus the build's batch size. */
typedef struct pdec_env_cfg {
  int pde_kind;            /* PDEC_PDE_* */
  int dtype;               /* PDEC_F32 / PDEC_F64 */
  int B;                   /* trajectories in the batch (reference: 1) */
  int N;                   /* nx: cells per species */
  int n_species;           /* 1 (KS) or 2 (Keller-Segel: rows u,v of y[2,nx]) */
  int S;                   /* number of sensors */
  int A;                   /* number of actuators (columns of state/action) */
  int window;              /* window_size (odd; rows per species = window) */
  int temporal_steps;      /* KellerSegelSetup.jl:48 */
  int mono;                /* 1: global agent (KSglobalSetup.jl): state [S,1], reward [1] */
  int K;                   /* oversampling: CNAB2 sub-steps / fixed RK4 sub-steps per control step */
  int check_max_value;     /* 0 none, 1 "y", 2 "reward" (src/PDEenv.jl:226-240) */
  double Lx;               /* domain length */
  double dt;               /* control interval */
  double mu;               /* KS disturbance amplitude (KSSetup.jl:155); 0 for mono */
  double max_value;        /* blow-up threshold */
  double sensor_scale;     /* sensors = <y,g> * sensor_scale  (1/max_value KS; 1/4 K-S) */
  double agent_power;      /* p = sum_i agent_power * action[i] * kernel_i */
  double reward_in_scale;  /* d = reward_in_scale * (<y,g> + reward_offset*sum(g)) */
  double reward_offset;    /* KS 0; K-S -1 (y-1) */
  double reward_power;     /* r = -|d|^reward_power / reward_denom - ... */
  double reward_denom;
  double action_punish;
  double delta_action_punish;
  /* 2-D fluid (PDEC_PDE_FLUID_RK4) only; ignored by the 1-D kinds.  Square periodic box: N = nx = ny,
   * Lx = Ly; y is the vorticity SPECTRUM, Julia ComplexF64[ny, nx] -> memory [B][nx][ny][re,im]. */
  int ifpad;               /* 1: 3/2-rule de-aliasing (scripts/Fluid/setup/FluidSetup.jl:101) */
  int sensors_per_axis;    /* sensors on a spa x spa grid, S = spa^2 (FluidSetup.jl:61) */
  double nu;               /* viscosity (FluidSetup.jl:28) */
  /* 2-D Keller-Segel (PDEC_PDE_KSEG2D_RK4) only: rows of the grid; N = nx (multiple of 4), square cells dx = Lx/nx */
  int Ny;
  /* time integrator of the right-hand-side kinds (PDEC_PDE_KSEG_RK4, PDEC_PDE_KS_RK4_FD): 0 = classical RK4 (what the
   * setups' do_step use), 1 = PDEenv's built-in explicit midpoint rule, K = `oversampling` sub-steps
   * (src/PDEenv.jl:208-214, taken when no do_step closure is supplied) */
  int integrator;
  /* action memory (scripts/KS/setup/KSSetup.jl:39,48,216-226; src/PDEagent.jl:201; 0 in every shipped script): the actor has
   * 1 + memory_size outputs per actuator; row 0 drives the PDE and the reward, rows 1.. come back as the LAST memory_size rows
   * of the next state (zeros at a reset).  action arrays become [B][A][1 + memory_size], state columns grow by memory_size.
   * Per-actuator kinds of the reference (KS CNAB2, KS RK4+FD, Keller-Segel, fluid); PDEC_E_INVALID for the global agent
 * (mono) and the 2-D Keller-Segel grid. */
  int memory_size;
} pdec_env_cfg;

/* sensor_kernels [S][N], actuator_kernels [A][N] (host, double, row = one kernel: the
 * `gaussians` / `gaussians_actuators` arrays, KSSetup.jl:111-113); a2s [A] 0-based index
 * of the sensor under each actuator (`actuators_to_sensors`).  Replaces the closures built
 * in KSSetup.jl:82-245 / KellerSegelSetup.jl:112-332. */
int pdec_env_create(pdec_handle* h, const pdec_env_cfg* cfg, const double* sensor_kernels,
                    const double* actuator_kernels, const int32_t* a2s);

/* 2-D fluid environment (src/fluid_rk4.jl:122-229 + scripts/Fluid/setup/FluidSetup.jl:139-261).
 * The sensor / actuator kernels are the thresholded Taylor-vortex bumps the reference keeps
 * `sparse` (FluidSetup.jl:139-161); each is passed as the dense BW x BH box around its periodic
 * support: boxes [S][BW][BH] (element (dj, di) at dj*BH + di; di runs along y = the fast axis of
 * the Julia array), origin [S][2] = {j0 (first x column), i0 (first y row)}, 0-based, wrapping
 * periodically; likewise for the A actuators.  The compute calls below then take
 *   y, p   : [B][nx][ny][2]  (spectra; p = fft(forcing), what prepare_action returns, :260)
 *   state  : [B][A][9*temporal_steps], reward [B][A], action [B][A]. */
int pdec_fluid_env_create(pdec_handle* h, const pdec_env_cfg* cfg, int BH, int BW,
                          const double* sensor_boxes, const int32_t* sensor_origin,
                          const double* actuator_boxes, const int32_t* actuator_origin,
                          const int32_t* a2s);

/* 2-D Keller-Segel environment (BASELINE.json configs[3]).  Sensors sit on the tensor grid
 * sensor_y[Sy] x sensor_x[Sx] (0-based cell indices of the box centres; sensor s = iy*Sx + ix), every sensor
 * and actuator kernel is the (2*half_window+1)^2 box of ones of KellerSegelSetup.jl:112-126, clipped at the
 * domain edge; a2s [A] as above (actuator boxes must not overlap).  cfg->N = nx, cfg->Ny = ny = the `ny` argument.
 *   y [B][ny][nx][2] (u,v interleaved; Julia y[2, nx, ny]), p [B][ny][nx], state [B][A][2*window^2*temporal_steps]
 * Feature rows: species u then v (:281-286), inside a species the (i,j) shifts of the 3x3 circular window in the
 * order of scripts/Fluid/setup/FluidSetup.jl:219-223, then the temporal stack (:297-303). */
int pdec_kseg2d_env_create(pdec_handle* h, const pdec_env_cfg* cfg, int ny, int Sx, int Sy,
                           const int32_t* sensor_x, const int32_t* sensor_y, int half_window,
                           const int32_t* a2s);

/* Fluid episode initialiser ic(caseno) (src/fluid_rk4.jl:72-120; taylorvtx :54-69) on the device: y_out[B][nx][ny][2]
 * = fft2 of the sum over nv vortices (9 periodic images each).  vortices: HOST [B][nv][4] = (x0, y0, a0, U_max), the
 * random draws of ic(3)/ic(4) made by the caller (scripts/Fluid/setup/FluidSetup.jl:386-394 generate_random_init). */
int pdec_fluid_ic(pdec_handle h, const double* vortices, int nv, void* y_out);

/* prepare_action(; env): p[B][N] from action[B][A]      (KSSetup.jl:231-245) */
int pdec_actuate(pdec_handle h, const void* action, void* p_out);
/* do_step(env): y_out[B][n_species*N] from y_in, p[B][N]; done[B] int32 (bit0 = blow-up,
 * only when check_max_value==1)                           (KSSetup.jl:130-160, PDEenv.jl:216-228) */
int pdec_pde_step(pdec_handle h, const void* y_in, const void* p, void* y_out, int32_t* done);
/* featurize(; env): state_out[B][A][ns] from y and the previous state (temporal_steps>1;
 * may be NULL = constructor/reset form, KSSetup.jl:190-229, KellerSegelSetup.jl:265-316) */
int pdec_featurize(pdec_handle h, const void* y, const void* prev_state, void* state_out);
/* the same with the action the environment holds (cfg.memory_size > 0: rows 1.. of `action` [B][A][1 + memory_size] become the
 * last memory_size rows of every state column, KSSetup.jl:220-226; action NULL = pdec_featurize = the reset form: zeros) */
int pdec_featurize_action(pdec_handle h, const void* y, const void* prev_state, const void* action, void* state_out);
/* reward_function(env): r_out[B][A] (or [B][1] mono)      (KSSetup.jl:162-178) */
int pdec_reward(pdec_handle h, const void* y, const void* action, const void* action_prev,
                void* r_out);
/* (env::PDEenv)(action), src/PDEenv.jl:195-241, in ONE launch: delta_action, prepare_action,
 * integrator, reward, featurize, blow-up flag.  y_out may alias y_in; state_out must not
 * alias state_prev when temporal_steps>1.  p_out may be NULL. */
int pdec_env_step(pdec_handle h, const void* y_in, const void* action, const void* action_prev,
                  const void* state_prev, void* y_out, void* p_out, void* state_out,
                  void* reward_out, int32_t* done);
/* Optional extra output of pdec_env_step for the batched DDPG update: terminal_per_column
 * [B][A] (mono: [B][1]) of the plan's dtype, 1.0 on every actuator column of a trajectory that
 * blew up in this step (the `terminal` trace RL.jl fills per actuator, src/PDEagent.jl:284-288),
 * 0.0 otherwise.  NULL (default) disables it.  The pointer is read at each later pdec_env_step. */
int pdec_env_set_terminal_out(pdec_handle h, void* terminal_per_column);
/* RHS evaluation for known-answer tests: out = f(y, p)    (KellerSegelSetup.jl:213-232) */
int pdec_rhs_eval(pdec_handle h, const void* y, const void* p, void* out);

/* Host-pointer forms (synchronous) -- what Julia's do_step/featurize closures bind. */
int pdec_pde_step_host(pdec_handle h, const void* y_in, const void* p, void* y_out, int32_t* done);
int pdec_env_step_host(pdec_handle h, const void* y_in, const void* action, const void* action_prev,
                       const void* state_prev, void* y_out, void* p_out, void* state_out,
                       void* reward_out, int32_t* done);

/* ---------------------------------------------------------------- networks ----------- */
/* Chain(Dense...) with weights shared across columns (src/PDEagent.jl:14-56).
 * dims[n_layers+1], acts[n_layers].  params_host: Flux.params order W1,b1,W2,b2,... each W
 * in Julia column-major [out,in] (element (o,i) at i*out+o), dtype = the plan's dtype; NULL
 * = zeros.  max_cols = largest number of columns any later call will pass. */
int pdec_mlp_create(pdec_handle* h, int dtype, int n_layers, const int32_t* dims,
                    const int32_t* acts, const void* params_host, int max_cols);
int pdec_mlp_num_params(pdec_handle h, int* n);
int pdec_mlp_set_params(pdec_handle h, const void* params_host);   /* Flux.loadparams!  (custom_nna.jl:26-27) */
int pdec_mlp_get_params(pdec_handle h, void* params_host);         /* Flux.params -> host (checkpoint)      */
int pdec_mlp_copy(pdec_handle dst, pdec_handle src);               /* copyto!(dst, src)  (custom_nna.jl:26)  */
/* app(x): x [cols][in] (= Julia [in, cols]), y_out [cols][out]    (custom_nna.jl:13) */
int pdec_mlp_forward(pdec_handle h, const void* x, int cols, void* y_out);
/* forward + backward of sum(dy .* app(x)): grads_out (device, layout of get_params, may be
 * NULL -> only the internal gradient buffer is filled), dx_out [cols][in] may be NULL */
int pdec_mlp_backward(pdec_handle h, const void* x, const void* dy, int cols, void* dx_out,
                      void* grads_out);
/* device pointer + length of the internal flat gradient buffer (for an external all-reduce) */
int pdec_mlp_grad_buffer(pdec_handle h, void** dptr, int* n);
/* update!(app, gs) with Flux.Optimise.ADAM semantics (custom_nna.jl:23-24); uses the
 * internal gradient buffer */
int pdec_adam_step(pdec_handle h, double eta, double beta1, double beta2, double eps);
int pdec_adam_get_state(pdec_handle h, void* m_host, void* v_host, double* beta_pow2);
int pdec_adam_set_state(pdec_handle h, const void* m_host, const void* v_host, const double* beta_pow2);
/* dest .= rho .* dest .+ (1-rho) .* src                          (src/PDEagent.jl:415-417)
 * Every `rho` of this header: the reference's loop runs over Flux.params([At, Ct]), which is empty (src/custom_nna.jl:20 defines a
 * functor of its own), so its targets never move; a caller reproduces the reference AS IT RUNS with rho = 1 (dest = 1*dest +
 * 0*src, exact) and the loop as written with rho = policy.p. */
int pdec_polyak(pdec_handle dst, pdec_handle src, double rho);

/* update!(app, gs) immediately followed by the Polyak step of the app's target network
 * (src/custom_nna.jl:23-24 + src/PDEagent.jl:415-417 for one network pair) in ONE launch; gradients are
 * taken from the internal gradient buffer (e.g. after pdec_allreduce_grads).  Polyak of a target only
 * depends on its own behaviour network, so doing it right after that network's ADAM step is equivalent
 * to the reference's order. */
int pdec_adam_polyak_step(pdec_handle h, pdec_handle h_target, double eta, double beta1, double beta2,
                          double eps, double rho);

/* T control steps in one call, no host round trip per step: for t = 0..T-1
 *   action_t = clamp(actor(state_t) + randn * act_noise, +-act_limit)   (src/PDEagent.jl:183-207, pdec_policy_act_rng
 *              with noise offset `offset + t * ceil(cols*na/4)`; learning = 0 -> no noise)
 *   (y, state, reward, done) = (env::PDEenv)(action_t)                  (src/PDEenv.jl:195-241, pdec_env_step)
 * all enqueued on the environment's stream (the actor handle must use the same stream and dtype).  In/out device
 * arrays: y, state, action (in: the previous action, for delta_action; out: the last one).  Optional outputs (NULL to
 * skip): reward_sum [B][A] += every step's reward; log_y / log_p / log_action / log_reward: [T][...] rows of the
 * trajectory (what PDEhook logs per step, src/PDEhook.jl:54-62); done_any [B] = OR of the step flags, done_step [B] =
 * first step that raised a flag (-1: none).  The reference stops an episode at `done`; a rollout keeps integrating
 * (blown-up trajectories saturate to inf/NaN) and reports the step, the caller discards what follows it.
 * KS (CNAB2, per-actuator agents, temporal_steps = 1, actor of <= 3 Dense layers no wider than 32 with one output):
 * ONE persistent launch for all T steps -- a workgroup keeps its two trajectories in registers and their state / actions
 * in LDS between steps and evaluates the actor itself on the vector unit (k-ordered sums like the oracle; not the
 * summation order of the MFMA acting kernel, so fp32 actions agree with the per-step loop to ~1e-6, not bit for bit).
 * 1-D Keller-Segel (per-actuator agents, any temporal_steps, same actor limits; scripts/Keller-Segel/setup/
 * KellerSegelSetup.jl:213-332): likewise one launch, one workgroup per trajectory.
 * PDEC_ROLLOUT_PERSISTENT=0 selects the per-step form, which every other configuration uses (same kernels as the loop
 * pdec_policy_act_rng -> pdec_env_step, bit-identical to it). */
int pdec_rollout(pdec_handle env, pdec_handle actor, int T, void* y, void* state, void* action,
                 double act_noise, double act_limit, int learning, uint64_t seed, uint64_t offset,
                 void* reward_sum, void* log_y, void* log_p, void* log_action, void* log_reward,
                 int32_t* done_any, int32_t* done_step);

/* policy act: actions[cols][na] = clamp(actor(state) + noise*act_noise, +-act_limit)
 * (src/PDEagent.jl:183-207).  noise [cols][na] device standard normals or NULL (-> no noise,
 * `learning=false`). */
int pdec_policy_act(pdec_handle actor, const void* state, const void* noise, int cols,
                    double act_noise, double act_limit, void* actions_out);
/* the same with the exploration noise drawn inside the kernel from the counter-based generator of
 * pdec_randn (identical numbers: element i of the stream (seed, offset) belongs to column i);
 * learning = 0 -> no noise (`learning=false`, src/PDEagent.jl:199).  One launch for 3-layer fp32 actors. */
/* exploration noise on the first `rows` outputs of the actor only (-1 = all, the default): with action memory the reference
 * adds noise to the driving row and leaves the memory rows as the actor produced them (src/PDEagent.jl:201:
 * actions[1:end-memory_size, :] += randn * act_noise); the clamp applies to every row.  Philox element numbering unchanged
 * (element = column * outputs + row; the memory rows' draws are skipped). */
int pdec_mlp_set_noise_rows(pdec_handle actor, int rows);
int pdec_policy_act_rng(pdec_handle actor, const void* state, int cols, double act_noise, double act_limit,
                        int learning, uint64_t seed, uint64_t offset, void* actions_out);
/* agent(env) where the environment computes in `state_dtype` and the actor keeps parameters of another type -- the reference's
 * case: fp64 fields, Float32 networks (src/PDEagent.jl:183-207 promotes in the broadcast) -- without a promoted copy of the
 * actor: when the single-launch acting kernel for few columns covers the case (<= 4 layers, activations of both buffers within
 * 48 KB of LDS) the action is enqueued and *served = 1; otherwise nothing is enqueued, *served = 0, and the caller acts through a
 * clone of the actor in `state_dtype` (pdec_mlp_copy + pdec_policy_act_rng).  Same arithmetic, bit for bit, either way. */
int pdec_policy_act_rng_as(pdec_handle actor, int state_dtype, const void* state, int cols, double act_noise, double act_limit,
                           int learning, uint64_t seed, uint64_t offset, void* actions_out, int* served);
/* The glue between two control steps of a single-trajectory training loop in ONE launch (src/PDEagent.jl:276-289, :175-209,
 * :254-274 in the order RL.jl's run loop calls them): pdec_replay_push_rt of the step that just ran (n_rt == 0: none), then
 * agent(env) for the next step -- act_mode 1: pdec_policy_act_rng_as on `state` [cols][ns] of type `dtype` with learning = 1;
 * 2: the zero action of the start policy; 0: none --, then pdec_replay_push_sa of (state, that action) (n_sa == 0: none; the
 * action rows are zero when act_mode == 0).  The episode halt flag of `trajectory_handle` (pdec_set_episode_halt) is honoured
 * and raised as by the three calls.  *served = 0 and nothing enqueued when the case is not the single-workgroup one (the caller
 * then makes the three calls): the actor's and the trajectory handle's streams differ, the acting call would not be served by
 * pdec_policy_act_rng_as, or a push does not fit one block.  Same stores, bit for bit.  done_event (0: none): an event of
 * pdec_event_create that the launch carries as its own completion event -- what pdec_event_record right behind the call would
 * give, without the record's packet in the stream (cf. pdec_mlp_set_stop_event). */
int pdec_step_glue(pdec_handle actor, pdec_handle trajectory_handle, int dtype, const void* reward, const int32_t* done_flags,
                   int cols_per_traj, int force_terminal, void* reward_trace, void* terminal_trace, int64_t capacity,
                   int64_t start_rt, int64_t n_rt, int act_mode, const void* state, int cols, double act_noise, double act_limit,
                   uint64_t seed, uint64_t offset, void* actions_out, void* state_trace, void* action_trace,
                   int64_t capacity_rows, int64_t start_sa, int64_t n_sa, pdec_handle done_event, int* served);
/* would pdec_step_glue serve (act_mode, cols columns of type dtype, a POST_ACT push of n_rt values)?  Nothing is enqueued. */
int pdec_step_glue_served(pdec_handle actor, pdec_handle trajectory_handle, int dtype, int act_mode, int cols, int64_t n_rt, int* served);
/* the same with the noise counter kept ON THE DEVICE (one per actor handle): the kernel reads the current counter and
 * one of its threads stores the advanced value (+ ceil(cols*na/4) when learning), so no launch argument depends on how
 * many calls came before -- the form a captured HIP graph of the control step replays (pdec_capture_begin).
 * pdec_noise_counter_set / _get move the counter (they synchronise the actor's stream). */
int pdec_policy_act_rng_dev(pdec_handle actor, const void* state, int cols, double act_noise, double act_limit,
                            int learning, uint64_t seed, void* actions_out);
/* 1 when the acting kernels of this actor read a PUBLISHED, double-buffered copy of its weights that a concurrent
 * update does not write (fp32 3-layer actors on the fused path); 0 when they read the parameters in place -- a caller that
 * overlaps acting and updating must then order the update's actor half behind the acting kernel. */
int pdec_mlp_acts_on_published_copy(pdec_handle actor, int* yes);
int pdec_noise_counter_set(pdec_handle actor, uint64_t value);
int pdec_noise_counter_get(pdec_handle actor, uint64_t* value);
/* fill dst[n] with standard normals from a counter-based generator (replaces randn(rng),
 * src/PDEagent.jl:201) */
int pdec_randn(pdec_handle any_handle, void* dst, size_t n, int dtype, uint64_t seed, uint64_t offset);

/* DDPG update (src/PDEagent.jl:363-418) split at the points where a data-parallel run
 * all-reduces gradients.  s,snext [Bu][ns]; a [Bu][na]; r [Bu]; t [Bu] (same dtype; 0/1).
 * quirk=1 reproduces the reference's (1xBu).+(Bu) broadcast of the reward,
 * quirk=0 is the diagonal TD target.  grad_scale multiplies the gradients (1/world_size for
 * a data-parallel mean).  Losses are written to device scalars (dtype of the plan). */
int pdec_ddpg_critic_grads(pdec_handle A, pdec_handle C, pdec_handle At, pdec_handle Ct,
                           const void* s, const void* a, const void* r, const void* t,
                           const void* snext, int Bu, double gamma, int quirk, double grad_scale,
                           void* critic_loss_dev);
int pdec_ddpg_actor_grads(pdec_handle A, pdec_handle C, const void* s, int Bu, double grad_scale,
                          void* actor_loss_dev);
/* whole update on one device: critic grads, ADAM(C), actor grads, ADAM(A), Polyak x2;
 * losses returned to host (synchronises). */
int pdec_ddpg_update(pdec_handle A, pdec_handle C, pdec_handle At, pdec_handle Ct,
                     const void* s, const void* a, const void* r, const void* t, const void* snext,
                     int Bu, double gamma, double rho, int quirk, double eta_actor, double eta_critic,
                     double* actor_loss, double* critic_loss);

/* whole update on one device WITHOUT host synchronisation: losses_dev[0] = critic loss, [1] = actor loss
 * (device scalars of the plan's dtype, may be NULL).  For 3-layer fp32 nets this is 4 launches: critic
 * pass, slab-reduce + ADAM(C) + Polyak(Ct), actor pass, slab-reduce + ADAM(A) + Polyak(At). */
int pdec_ddpg_update_async(pdec_handle A, pdec_handle C, pdec_handle At, pdec_handle Ct,
                           const void* s, const void* a, const void* r, const void* t, const void* snext,
                           int Bu, double gamma, double rho, int quirk, double eta_actor, double eta_critic,
                           void* losses_dev);

/* The reference's actual update shape -- `update_loops` updates per control step on minibatches of `batch_size`
 * transitions (src/PDEagent.jl:342-418; KSSetup.jl:66-71: 20 x 3) -- in ONE launch: minibatch gather from the
 * device-resident replay traces (pde_fetch!, :323-340), the whole update, repeated `loops` times by one workgroup.
 * Traces (fp32, as RL.jl keeps them, :112-117): state [slots][ns], action [slots][na], reward [slots],
 * terminal [slots] (0/1).  idx_s / idx_rt / idx_sn: device int32 [loops][Bu] slots of (s,a), (r,t) and s',
 * drawn by the host as pde_sample does (:317-321).  fp32 networks of up to 4 layers, 1 <= Bu <= 16.
 * losses_dev (may be NULL): [critic, actor] loss of the last loop.  No host synchronisation. */
int pdec_ddpg_update_small(pdec_handle A, pdec_handle C, pdec_handle At, pdec_handle Ct,
                           const void* state_trace, const void* action_trace, const void* reward_trace,
                           const void* terminal_trace, const int32_t* idx_s, const int32_t* idx_rt,
                           const int32_t* idx_sn, int loops, int Bu, double gamma, double rho, int quirk,
                           double eta_actor, double eta_critic, void* losses_dev);

/* The batch-mean reward of the reference's (1 x Bu) .+ (Bu) broadcast (quirk = 1) reduced AHEAD of the
 * update: pdec_reward_mean(h, r, n, mean_out) sums the fp32 rewards r [n] in a fixed order on h's stream (the producer of
 * r -- the env step -- does it beside the running update) and pdec_ddpg_set_reward_mean hands the device scalar to the
 * NEXT critic pass of `critic`, which then does not read r for the mean (one hand-over per update; fp32 fused paths). */
int pdec_reward_mean(pdec_handle any_handle, const void* r, int n, void* mean_out);
/* For callers that run the fused KS step beside the f32-MFMA update passes on a second stream (the two-stream training
 * step): launch the step in the form that needs <= 64 VGPRs -- per-mode constants in LDS instead of registers -- so that a
 * wave of it can share a SIMD with two waves of the 222-VGPR critic pass instead of excluding a whole workgroup of it
 * (and being excluded by it) per CU; that form also runs at wave priority 3.  Alone it is slower (37 vs 29 us at C2), so
 * it is off by default.  *effective = 1 when the environment has such a form (KS CNAB2, N = 256, fp32), else 0. */
int pdec_env_set_simd_sharing(pdec_handle env, int on, int* effective);
/* the same without any extra launch for the fused KS steps + 3-layer fused critic: every later pdec_env_step also writes the
 * sum of the rewards of each of its workgroups (CNAB2: two trajectories each, RK4 + FD: one) to partial_sums [*n_partials]
 * (fp32; NULL switches it off), and pdec_ddpg_set_reward_partials hands them to the next critic pass, which adds them in a
 * fixed order. */
int pdec_env_set_reward_partials_out(pdec_handle env, void* partial_sums, int* n_partials);
/* Environments that run parts of their batch on streams of their own (2-D Keller-Segel: the RK4 sub-steps of C4 in three
 * parts; fluid: child environments) make those streams themselves by default.  pdec_env_part_streams tells how many the step
 * uses besides the environment's own; pdec_env_set_part_streams hands over the caller's instead (made back to back with its
 * pipeline streams, see pdec_stream_create; the library's are released, the caller's are never destroyed by the library).
 * Fewer than asked for: 2-D Keller-Segel runs that many parts + 1 (0: unsplit), fluid refuses (PDEC_E_INVALID).  Environments
 * without parts accept and ignore the call.  Same results bit for bit either way.
 * The library's own part streams (and their events) are made at the FIRST split step, not at environment creation -- by then
 * the environment's stream, whose priority level they take, is known.  Nothing can be created while a stream is being
 * captured, so a caller that opens pdec_capture_begin before any eager step must either run one step first or hand over its
 * own streams with pdec_env_set_part_streams; otherwise the captured step returns PDEC_E_INVALID naming this. */
int pdec_env_part_streams(pdec_handle env, int* n);
int pdec_env_set_part_streams(pdec_handle env, void* const* hip_streams, int n);
int pdec_ddpg_set_reward_partials(pdec_handle critic, const void* partial_sums, int n);
int pdec_ddpg_set_reward_mean(pdec_handle critic, const void* mean_dev);

/* the same with pde_sample (src/PDEagent.jl:317-321) INSIDE the kernel: draw k = loop * Bu + column is word k % 4 of the
 * Philox block (seed; counter offset + k / 4), ind = (word * (n_valid - stride)) >> 32, logical index
 * max(0, n_rt - capacity) + ind, slots (lg mod (capacity + stride), lg mod capacity, (lg + stride) mod (capacity + stride)).
 * n_valid = length(trajectory) = min(n_rt, capacity), n_rt = entries ever pushed to the reward / terminal traces,
 * stride = number of columns one control step pushes (number_actuators x batch).  The host passes a seed and an
 * offset, no index arrays. */
int pdec_ddpg_update_small_rng(pdec_handle A, pdec_handle C, pdec_handle At, pdec_handle Ct,
                               const void* state_trace, const void* action_trace, const void* reward_trace,
                               const void* terminal_trace, int loops, int Bu, uint64_t seed, uint64_t offset,
                               int64_t n_valid, int64_t n_rt, int64_t capacity, int stride, double gamma, double rho,
                               int quirk, double eta_actor, double eta_critic, void* losses_dev);

/* the two halves of pdec_ddpg_update_async as separate calls (critic half: critic pass + ADAM(C) + Polyak(Ct);
 * actor half: actor pass with the updated critic + ADAM(A) + Polyak(At)), so that a caller can order the
 * actor half behind a concurrent reader of the actor's weights on another stream.  losses_dev as above. */
int pdec_ddpg_update_critic_async(pdec_handle A, pdec_handle C, pdec_handle At, pdec_handle Ct,
                                  const void* s, const void* a, const void* r, const void* t, const void* snext,
                                  int Bu, double gamma, double rho, int quirk, double eta_critic, void* losses_dev);
int pdec_ddpg_update_actor_async(pdec_handle A, pdec_handle C, pdec_handle At, pdec_handle Ct,
                                 const void* s, int Bu, double rho, double eta_actor, void* losses_dev);

/* ---------------------------------------------------------------- replay ---- */
/* The trajectory glue of src/PDEagent.jl:237-340 on device-resident traces (fp32, as RL.jl keeps them, :112-117):
 * state [capacity + stride][ns], action [capacity + stride][na], reward [capacity], terminal [capacity].  The host
 * keeps only the two entry counters (pop of the dummy (s, a) of an episode end, :237-252, is a counter decrement).
 * PRE_ACT push of one (s, a) column per actuator (:254-274): rows start .. start+n-1 (mod capacity_rows) of both traces
 * from s [n][ns], a [n][na] of `dtype`; a == NULL pushes zero actions (the POST_EPISODE dummy, :291-314).  One launch. */
int pdec_replay_push_sa(pdec_handle any_handle, void* state_trace, void* action_trace, int64_t capacity_rows, int ns, int na,
                        int64_t start, const void* s, const void* a, int64_t n, int dtype);
/* POST_ACT push of (r, terminal) (:276-289): r [n] of `dtype`; terminal of column i = done_flags[i / cols_per_traj] != 0
 * (the env step's per-trajectory flags), or 1 everywhere when force_terminal (time-out, src/PDEenv.jl:227).  One launch. */
int pdec_replay_push_rt(pdec_handle any_handle, void* reward_trace, void* terminal_trace, int64_t capacity_rows, int64_t start,
                        const void* r, const int32_t* done_flags, int cols_per_traj, int force_terminal, int64_t n, int dtype);
/* pde_sample + pde_fetch! (:317-340) for a batch of Bu transitions in one launch: indices from the Philox stream
 * (seed, offset) exactly as pdec_ddpg_update_small_rng draws them; outputs fp32 s, sn [Bu][ns], a [Bu][na], r, t [Bu];
 * slots_out (optional) int32 [3][Bu] = the slots used. */
int pdec_replay_sample(pdec_handle any_handle, const void* state_trace, const void* action_trace, const void* reward_trace,
                       const void* terminal_trace, int ns, int na, int64_t capacity, int stride, int64_t n_valid, int64_t n_rt,
                       uint64_t seed, uint64_t offset, int Bu, void* s_out, void* a_out, void* r_out, void* t_out,
                       void* sn_out, int32_t* slots_out);

/* Speculatively issued episodes (run.py: the control steps of a whole episode are enqueued without reading the environment's
 * `done` flag back after every step, src/PDEenv.jl:226-240 / RL.jl's `while !is_terminated(env)`).  `flag` is a device int32,
 * zero at the episode's start.  Attached to a handle, it makes the launches issued THROUGH that handle that change persistent
 * learner state no-ops once it is raised: pdec_replay_push_sa / pdec_replay_push_rt (the handle passed as any_handle) and
 * pdec_ddpg_update_small(_rng) (the behaviour critic's handle; the ADAM beta powers move on unchanged).  pdec_replay_push_rt
 * raises it after pushing a transition whose done_flags[0] != 0 (B = 1: the step that ends the episode is pushed, nothing
 * after it).  Everything else a later step writes goes to per-step buffers the host ignores.  NULL detaches. */
int pdec_set_episode_halt(pdec_handle any_handle, int32_t* flag);

/* Batched environments (B > 1; the reference has B = 1 and ends the episode at the first blow-up, src/PDEenv.jl:226-240):
 * same-step reset of every trajectory whose done flag is raised -- y, state and action rows are overwritten by their
 * initial images (reset!, :183-193), a non-finite reward becomes 0 -- so that a blown-up trajectory cannot feed inf/NaN
 * states into the learner for the rest of the lock-stepped episode.  state/state0, action/action0, reward may be NULL. */
int pdec_env_autoreset(pdec_handle env, const int32_t* done, void* y, const void* y0, void* state, const void* state0,
                       void* action, const void* action0, void* reward);
/* generate_random_init() of the 1-D setups (scripts/KS/setup/KSSetup.jl:288-298,
 * scripts/Keller-Segel/setup/KellerSegelSetup.jl:373-384) on the device, one workgroup per trajectory: y0_out [B][...] in
 * the environment's layout and dtype; uniforms from the Philox stream (seed, offset) (Julia's global RNG cannot be
 * reproduced, only the distribution; oracle/rng.py restates this stream).  Consumes B * ceil(n_coefficients / 4) counters. */
int pdec_env_random_init(pdec_handle env, uint64_t seed, uint64_t offset, void* y0_out);

/* ---------------------------------------------------------------- HIP graphs -- */
/* Record everything enqueued on the stream of `origin` (pdec_set_stream; not the null stream) -- library calls, and any
 * other stream that forks from it and joins it again through events -- between _begin and _end into one HIP graph;
 * pdec_graph_launch replays it with one host call.  Calls that keep their per-step state on the device
 * (pdec_policy_act_rng_dev, every ADAM step) replay correctly: the slot selectors of their double-buffered device
 * state, which the captured calls flipped on the host, are flipped again by every pdec_graph_launch after the first
 * (capturing records the work without running it, so the first launch is the execution the capture stands for).
 * A graph must be replayed at a step whose buffers are those of the captured one (same ring phase).
 * Precondition for environments that step their batch in parts (fluid, 2-D Keller-Segel): their part streams exist -- one
 * eager step or pdec_env_set_part_streams before the capture (see pdec_env_part_streams).
 * The graph handle is released with pdec_destroy. */
int pdec_capture_begin(pdec_handle origin);
int pdec_capture_end(pdec_handle origin, pdec_handle* graph);
int pdec_graph_launch(pdec_handle graph, void* hip_stream);      /* NULL = the stream it was captured on */
int pdec_graph_num_nodes(pdec_handle graph, int* n);
/* One-shot: the next slab-reduction / ADAM launch issued on `mlp` (the second kernel of pdec_ddpg_update_critic_async /
 * pdec_ddpg_update_actor_async on the fused 3-layer path) carries `event` as the completion event of its own dispatch
 * packet: what pdec_event_record(event, stream) right behind that call would give, without the record's own packet in the
 * stream (each costs the update chain ~4.5 us).  PDEC_E_INVALID for networks outside that path. */
int pdec_mlp_set_stop_event(pdec_handle mlp, pdec_handle event);
/* The same for the next reduce-ONLY launch on `mlp` (the second kernel of pdec_ddpg_actor_grads / pdec_ddpg_critic_grads on the
 * fused 3-layer path: the flat gradient buffer is complete when it ends) -- the event an all-reduce issued on ANOTHER stream
 * waits for, so that the collective leaves the update chain (pipeline.py).  A launch that applies the update never consumes
 * it, a reduce-only launch never consumes the stop event, and pdec_ddpg_actor_grads / _critic_grads record it behind their
 * last launch when the path they took has no such launch (so a waiter never sees a stale event). */
int pdec_mlp_set_reduce_event(pdec_handle mlp, pdec_handle event);
/* records a still pending stop event on the net's stream (the update took a path that does not consume it); else no-op */
int pdec_mlp_flush_stop_event(pdec_handle mlp);
/* Device-scope events for the hand-offs between two streams of one device (no system-scope cache fence, unlike a
 * default HIP event): record on one stream, make another stream wait.  Released with pdec_destroy. */
int pdec_event_create(pdec_handle* ev);
int pdec_event_record(pdec_handle ev, void* hip_stream);
int pdec_stream_wait_event(void* hip_stream, pdec_handle ev);

/* A device-side hand-over between two single-workgroup launches on DIFFERENT streams, for the NEXT launch through `handle`
 * that supports it -- pdec_step_glue (the actor's handle) and the fused fp64 KS step of one trajectory with 192 / 240 / 600 cells
 * in pdec_env_step (the environment's handle), the two launches that alternate on the chain of the reference-shaped training loop
 * (src/PDEenv.jl:195-241 and src/PDEagent.jl:175-289 in RL.jl's run order): the launch waits inside the kernel until
 * *wait_flag >= wait_value before it reads anything (NULL: no wait; the wait gives up after 0.3 s and counts a timeout) and
 * stores done_value to *done_flag behind its last store (NULL: no signal).  Flags: device int64, zeroed by the caller, values
 * increasing.  One-shot: cleared by the launch.  PDEC_E_INVALID from a launch that cannot honour a pending one. */
int pdec_set_launch_sync(pdec_handle handle, const int64_t* wait_flag, int64_t wait_value, int64_t* done_flag, int64_t done_value);
/* Do launches on the two streams run side by side?  HIP maps streams onto a few hardware queues; two streams that share one run
 * in issue order, and a pdec_set_launch_sync wait for a launch queued behind the waiter on the same hardware queue would wait for
 * nothing.  A caller checks its pair of streams with this once (a 20-ms bounded probe; synchronises both) before it uses
 * pdec_set_launch_sync across them.  *yes = 1: side by side. */
int pdec_streams_run_side_by_side(void* stream_a, void* stream_b, int* yes);
/* waits that gave up since the library was loaded (synchronous read) */
int pdec_launch_sync_timeouts(int* n);

/* ---------------------------------------------------------------- multi-GPU ---------- */
/* One RCCL communicator per process (one process per GPU).  unique_id: 128 bytes from
 * pdec_comm_unique_id on rank 0, broadcast by the host (the reference has no collective;
 * this is the build's data-parallel addition). */
int pdec_comm_unique_id(void* id128);
int pdec_comm_create(pdec_handle* c, int nranks, int rank, const void* id128);
/* the same with a bounded wait: RCCL's rendezvous blocks until every rank has arrived, so one missing rank would park the
 * others for ever; past timeout_ms (> 0) this returns PDEC_E_COMM instead and the caller can agree on another path
 * (bench.py: torch.distributed).  timeout_ms <= 0 = pdec_comm_create. */
int pdec_comm_create_timeout(pdec_handle* c, int nranks, int rank, const void* id128, int timeout_ms);
/* sum-all-reduce the internal gradient buffer of an MLP in place (fp32/fp64), on the network's stream ... */
int pdec_allreduce_grads(pdec_handle comm, pdec_handle mlp);
/* ... or on `hip_stream` (NULL = the network's): the caller orders it against the launch that leaves the gradient
 * (pdec_mlp_set_reduce_event) and the one that consumes it (pdec_stream_wait_event before pdec_adam_polyak_step) */
int pdec_allreduce_grads_on(pdec_handle comm, pdec_handle mlp, void* hip_stream);
int pdec_allreduce(pdec_handle comm, void* dptr, size_t n, int dtype, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* PDECONV_H */
