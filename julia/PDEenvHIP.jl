# PDEenvHIP.jl -- Julia binding of libpdeconv.so (include/pdeconv.h), the file a maintainer of
# janstenner/DistributedConvRL-PDE-Control drops into `src/` (see INTEGRATION.md).
#
# The reference has no FFI; its seam is the set of closures `PDEenv` calls (`do_step`, `featurize`,
# `prepare_action`, `reward_function`: src/PDEenv.jl:195-241) and the NN wrapper (src/custom_nna.jl:7-27,
# `RLBase.update!(policy, batch)` at src/PDEagent.jl:363).  Every function below is a thin `ccall`; nothing is
# computed in Julia.  Julia is not installed in the build image, so this file is NOT executed there: its struct
# mirror, the symbol names and the argument counts of every `ccall` are checked against include/pdeconv.h by
# tests/test_host_logic.py::test_julia_glue_matches_the_c_abi, and the identical C ABI is exercised by the Python
# ctypes host and the GPU tests.
module PDEenvHIP

using Flux
import ReinforcementLearning: RLBase

const LIB = get(ENV, "PDECONV_LIB", joinpath(@__DIR__, "..", "libpdeconv.so"))

const F32, F64 = Cint(0), Cint(1)
const KS_CNAB2, KSEG_RK4, KS_RK4_FD, FLUID_RK4, KSEG2D_RK4 = Cint(0), Cint(1), Cint(2), Cint(3), Cint(4)

# mirror of `struct pdec_env_cfg` (include/pdeconv.h): same fields, same order, same C types
struct EnvCfg
    pde_kind::Cint
    dtype::Cint
    B::Cint
    N::Cint
    n_species::Cint
    S::Cint
    A::Cint
    window::Cint
    temporal_steps::Cint
    mono::Cint
    K::Cint
    check_max_value::Cint
    Lx::Cdouble
    dt::Cdouble
    mu::Cdouble
    max_value::Cdouble
    sensor_scale::Cdouble
    agent_power::Cdouble
    reward_in_scale::Cdouble
    reward_offset::Cdouble
    reward_power::Cdouble
    reward_denom::Cdouble
    action_punish::Cdouble
    delta_action_punish::Cdouble
    ifpad::Cint
    sensors_per_axis::Cint
    nu::Cdouble
    Ny::Cint
    integrator::Cint
    memory_size::Cint
end

# keyword constructor: every field by name, so that a field added to the C struct cannot silently shift the rest
function EnvCfg(; pde_kind = KS_CNAB2, dtype = F64, B = 1, N, n_species = 1, S, A, window = 1, temporal_steps = 1,
                mono = 0, K, check_max_value = 1, Lx, dt, mu = 0.0, max_value, sensor_scale, agent_power,
                reward_in_scale, reward_offset = 0.0, reward_power, reward_denom, action_punish, delta_action_punish,
                ifpad = 0, sensors_per_axis = 0, nu = 0.0, Ny = 0, integrator = 0, memory_size = 0)
    EnvCfg(pde_kind, dtype, B, N, n_species, S, A, window, temporal_steps, mono, K, check_max_value, Lx, dt, mu,
           max_value, sensor_scale, agent_power, reward_in_scale, reward_offset, reward_power, reward_denom,
           action_punish, delta_action_punish, ifpad, sensors_per_axis, nu, Ny, integrator, memory_size)
end

check(rc) = rc == 0 || error(unsafe_string(ccall((:pdec_last_error, LIB), Cstring, ())))
init(dev = 0) = check(ccall((:pdec_init, LIB), Cint, (Cint,), dev))
shutdown() = check(ccall((:pdec_shutdown, LIB), Cint, ()))
destroy(h::UInt64) = check(ccall((:pdec_destroy, LIB), Cint, (UInt64,), h))

# ---- device memory helpers (pdec_malloc / pdec_memcpy_*)
function device_alloc(bytes::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pdec_malloc, LIB), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, bytes))
    p[]
end
device_free(p::Ptr{Cvoid}) = check(ccall((:pdec_free, LIB), Cint, (Ptr{Cvoid},), p))
function device_upload(a::Array)
    p = device_alloc(sizeof(a))
    check(ccall((:pdec_memcpy_h2d, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), p, a, sizeof(a)))
    p
end
device_download!(a::Array, p::Ptr{Cvoid}) =
    (check(ccall((:pdec_memcpy_d2h, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), a, p, sizeof(a))); a)

# ---- environment ------------------------------------------------------------------------------------------------
# gaussians :: Vector{Vector{Float64}} as built by prepare_gaussians (scripts/KS/setup/KSSetup.jl:111-113)
function env_create(cfg::EnvCfg, gaussians, gaussians_actuators, actuators_to_sensors)
    h = Ref{UInt64}(0)
    G = collect(reduce(hcat, gaussians))                 # column-major [N, S] == row-major [S][N]
    Ga = collect(reduce(hcat, gaussians_actuators))
    a2s = Int32.(actuators_to_sensors .- 1)
    check(ccall((:pdec_env_create, LIB), Cint,
                (Ref{UInt64}, Ref{EnvCfg}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}), h, cfg, G, Ga, a2s))
    note_experiment_family!(cfg.pde_kind)
    h[]
end

# the configuration of scripts/KS/setup/KSSetup.jl from its globals
ks_cfg(; nx, Lx, dt, oversampling, mu, max_value, agent_power, action_punish, delta_action_punish, n_sensors,
       n_actuators, window_size = 1, temporal_steps = 1, B = 1, dtype = F64) =
    EnvCfg(pde_kind = KS_CNAB2, dtype = dtype, B = B, N = nx, S = n_sensors, A = n_actuators, window = window_size,
           temporal_steps = temporal_steps, K = oversampling, Lx = Lx, dt = dt, mu = mu, max_value = max_value,
           sensor_scale = 1 / max_value, agent_power = agent_power, reward_in_scale = 6.0, reward_power = 1.3,
           reward_denom = 3 * max_value, action_punish = action_punish, delta_action_punish = delta_action_punish)

# do_step(env) -> y_new                               replaces scripts/KS/setup/KSSetup.jl:130-160
function do_step(h::UInt64, y::Array{Float64}, p::Array{Float64})
    ynew = similar(y)
    done = Ref{Int32}(0)
    check(ccall((:pdec_pde_step_host, LIB), Cint,
                (UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}), h, y, p, ynew, done))
    ynew
end

# the whole (env::PDEenv)(action) body in one launch    replaces src/PDEenv.jl:196-222
function env_step!(h::UInt64, env, action)
    ynew = similar(env.y)
    p = zeros(size(env.y)[end])
    state = similar(env.state)
    reward = zeros(size(env.state, 2))
    done = Ref{Int32}(0)
    check(ccall((:pdec_env_step_host, LIB), Cint,
                (UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Ref{Int32}),
                h, env.y, action, env.action, env.state, ynew, p, state, reward, done))
    ynew, p, state, reward, done[] != 0
end

# T acting-only control steps -- action = clamp(actor(state) + randn * act_noise), (env::PDEenv)(action) -- in ONE call
# (evaluation episodes / data collection: src/PDEagent.jl:175-209 + src/PDEenv.jl:195-241 without the host in the loop;
# KS: one persistent launch).  y, state, action: device arrays (device_upload) updated in place; reward_sum [B][A]
# accumulates.  The caller synchronises afterwards (pdec_sync).
function rollout!(h::UInt64, actor, T::Integer, y::Ptr{Cvoid}, state::Ptr{Cvoid}, action::Ptr{Cvoid};
                  act_noise = 0.0, act_limit = 1.0, learning = false, seed = 0, offset = 0, reward_sum = C_NULL)
    check(ccall((:pdec_rollout, LIB), Cint,
                (UInt64, UInt64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, UInt64, UInt64, Ptr{Cvoid},
                 Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}),
                h, actor.h, T, y, state, action, act_noise, act_limit, learning ? 1 : 0, seed, offset, reward_sum,
                C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
end

# for callers that run the fused KS step beside the update passes on a second stream: its 64-VGPR form (see pdeconv.h)
function set_simd_sharing(h::UInt64, on::Bool)
    eff = Ref{Cint}(0)
    check(ccall((:pdec_env_set_simd_sharing, LIB), Cint, (UInt64, Cint, Ref{Cint}), h, on ? 1 : 0, eff))
    eff[] != 0
end

# 2-D fluid: gaussians[i] are the `sparse` thresholded bumps of scripts/Fluid/setup/FluidSetup.jl:139-161, passed as
# the dense BW x BH box around the periodic support of each (boxes [S][BW][BH], origin (j0, i0), 0-based)
function fluid_env_create(cfg::EnvCfg, BH, BW, sensor_boxes, sensor_origin, actuator_boxes, actuator_origin, a2s)
    h = Ref{UInt64}(0)
    check(ccall((:pdec_fluid_env_create, LIB), Cint,
                (Ref{UInt64}, Ref{EnvCfg}, Cint, Cint, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}),
                h, cfg, BH, BW, sensor_boxes, sensor_origin, actuator_boxes, actuator_origin, a2s))
    note_experiment_family!(cfg.pde_kind)
    h[]
end
# do_step(env) (FluidSetup.jl:163-172): env.y, env.p are ComplexF64[ny, nx], passed as they lie (re, im interleaved)
function fluid_do_step(h::UInt64, y::Matrix{ComplexF64}, p::Matrix{ComplexF64})
    ynew = similar(y)
    check(ccall((:pdec_pde_step_host, LIB), Cint, (UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}),
                h, y, p, ynew, C_NULL))
    ynew
end
# generate_random_init() (FluidSetup.jl:386-394 -> ic(3)/ic(4), src/fluid_rk4.jl:72-120): the random draws stay in
# Julia, the 9-image Taylor-vortex sums and the fft2 run on the GPU.  vort: 4 x nv (x0, y0, a0, U_max), y0_dev: device
fluid_ic!(h::UInt64, vort::Matrix{Float64}, y0_dev::Ptr{Cvoid}) =
    check(ccall((:pdec_fluid_ic, LIB), Cint, (UInt64, Ptr{Cdouble}, Cint, Ptr{Cvoid}), h, vort, size(vort, 2), y0_dev))

# Keller-Segel on a 2-D grid (BASELINE.json configs[3]; the reference script is 1-D): sensor centres on a tensor grid
# (0-based cells), (2 half_window + 1)^2 boxes of ones as in prepare_rectangles (KellerSegelSetup.jl:112-126)
function kseg2d_env_create(cfg::EnvCfg, ny, sensor_x::Vector{Int32}, sensor_y::Vector{Int32}, half_window, a2s::Vector{Int32})
    h = Ref{UInt64}(0)
    check(ccall((:pdec_kseg2d_env_create, LIB), Cint,
                (Ref{UInt64}, Ref{EnvCfg}, Cint, Cint, Cint, Ptr{Int32}, Ptr{Int32}, Cint, Ptr{Int32}),
                h, cfg, ny, length(sensor_x), length(sensor_y), sensor_x, sensor_y, half_window, a2s))
    note_experiment_family!(cfg.pde_kind)
    h[]
end

# ---- the NN seam: a model type usable inside CustomNeuralNetworkApproximator ---------------------------------------
mutable struct HipMLP
    h::UInt64
    dims::Vector{Int32}
    acts::Vector{Int32}
end
nparams(m::HipMLP) = sum(m.dims[i] * m.dims[i + 1] + m.dims[i + 1] for i in 1:length(m.acts))
function HipMLP(chain::Flux.Chain; max_cols = 1)        # upload an existing Flux Chain(Dense...)
    dims = Int32[size(chain[1].weight, 2); [size(l.weight, 1) for l in chain]...]
    acts = Int32[l.σ === relu ? 1 : l.σ === tanh ? 2 : 0 for l in chain]
    flat = reduce(vcat, [vcat(vec(l.weight), l.bias) for l in chain])      # Flux.params order, W column-major
    h = Ref{UInt64}(0)
    check(ccall((:pdec_mlp_create, LIB), Cint, (Ref{UInt64}, Cint, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Cvoid}, Cint),
                h, F32, length(acts), dims, acts, Float32.(flat), max_cols))
    HipMLP(h[], dims, acts)
end
function unflatten(m::HipMLP, flat::Vector{Float32})
    out, o = Any[], 0
    for i in 1:length(m.acts)
        nin, nout = m.dims[i], m.dims[i + 1]
        push!(out, reshape(flat[o+1:o+nin*nout], Int(nout), Int(nin))); o += nin * nout
        push!(out, flat[o+1:o+nout]); o += nout
    end
    out
end
function (m::HipMLP)(x::AbstractMatrix)                 # app(x), src/custom_nna.jl:13; x is [in, cols]
    xs = Float32.(x)
    cols = size(xs, 2)
    y = Matrix{Float32}(undef, m.dims[end], cols)
    dx = device_upload(xs)
    dy = device_alloc(sizeof(y))
    check(ccall((:pdec_mlp_forward, LIB), Cint, (UInt64, Ptr{Cvoid}, Cint, Ptr{Cvoid}), m.h, dx, cols, dy))
    device_download!(y, dy)
    device_free(dx); device_free(dy)
    y
end
function Flux.params(m::HipMLP)
    flat = Vector{Float32}(undef, nparams(m))
    check(ccall((:pdec_mlp_get_params, LIB), Cint, (UInt64, Ptr{Cvoid}), m.h, flat))
    unflatten(m, flat)
end
Base.copyto!(dst::HipMLP, src::HipMLP) = check(ccall((:pdec_mlp_copy, LIB), Cint, (UInt64, UInt64), dst.h, src.h))
# memory_size > 0 (EnvCfg(memory_size = m); actions [1 + m, A]): exploration noise on the first `rows` actor outputs only,
# as actions[1:end-memory_size, :] += randn * act_noise does (src/PDEagent.jl:201)
set_noise_rows!(m::HipMLP, rows::Integer) = check(ccall((:pdec_mlp_set_noise_rows, LIB), Cint, (UInt64, Cint), m.h, rows))
function Base.deepcopy(m::HipMLP)                       # PDEhook snapshots the actor (src/PDEhook.jl:37-38)
    h = Ref{UInt64}(0)
    check(ccall((:pdec_mlp_create, LIB), Cint, (Ref{UInt64}, Cint, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Cvoid}, Cint),
                h, F32, length(m.acts), m.dims, m.acts, C_NULL, 1))
    c = HipMLP(h[], copy(m.dims), copy(m.acts))
    copyto!(c, m)
    c
end

# RLBase.update!(policy, batch) -- Zygote cannot differentiate through ccall, so the method at src/PDEagent.jl:363 is
# overridden with the fused call (same losses, ADAM, Polyak).  `CustomDDPGPolicy` / `CustomNeuralNetworkApproximator`
# are the reference's own types (src/PDEagent.jl:121, src/custom_nna.jl:7), defined in Main before this file is included.
# FROZEN_TARGETS: the reference's Polyak loop (src/PDEagent.jl:415-417) iterates over Flux.params([At, Ct]), which is EMPTY --
# src/custom_nna.jl:20 defines a `functor` of its own instead of extending Functors.functor -- so its target networks never
# move (scripts/KS/KS22/saves/agent.jld2: zero target biases after 130 340 updates).  true = the reference as it runs (the
# kernels get rho = 1: dest = 1 * dest + 0 * src); false = the loop as written (rho = policy.p).
# The default is PER EXPERIMENT FAMILY, as in the Python host (setup.reproduces_reference_with): creating an environment sets it
# to the regime under which this path reproduces the reference's saved runs of that family -- frozen for KS (KS22 / KS200),
# moving for Keller-Segel and the fluid, whose artifacts a later session of the authors wrote (under frozen targets
# Keller-Segel saturates at return -30 in 24 of 24 seeds and the fluid diverges in 4 of 6: HISTORY.md 5.1).
# `set_target_networks!(:frozen | :moving)` pins a choice; a pinned choice that differs from the family's regime is honoured
# with a warning.
const FROZEN_TARGETS = Ref(true)
const TARGETS_PINNED = Ref(false)
const REPRODUCES_REFERENCE_WITH = Dict(KS_CNAB2 => :frozen, KS_RK4_FD => :frozen, KSEG_RK4 => :moving, KSEG2D_RK4 => :moving,
                                       FLUID_RK4 => :moving)
function set_target_networks!(regime::Symbol)
    regime in (:frozen, :moving) || error("set_target_networks!: :frozen or :moving")
    FROZEN_TARGETS[] = regime == :frozen
    TARGETS_PINNED[] = true
end
function note_experiment_family!(pde_kind)
    want = REPRODUCES_REFERENCE_WITH[Cint(pde_kind)]
    if !TARGETS_PINNED[]
        FROZEN_TARGETS[] = want == :frozen
    elseif FROZEN_TARGETS[] != (want == :frozen)
        @warn "target networks pinned to $(FROZEN_TARGETS[] ? :frozen : :moving), but the reference's saved runs of this experiment family are reproduced only with $want targets (HISTORY.md 5.1)"
    end
end
function ddpg_update!(policy, batch)
    s, a, r, t, snext = batch                            # Float32; s [ns,Bu], a [na,Bu], r [1,Bu], t [Bu]
    rho = FROZEN_TARGETS[] ? 1.0 : Float64(policy.p)
    al = Ref{Cdouble}(0)
    cl = Ref{Cdouble}(0)
    Bu = size(s, 2)
    ds, da, dr, dt, dsn = device_upload.((Array(s), Array(a), vec(Array(r)), Float32.(t), Array(snext)))
    check(ccall((:pdec_ddpg_update, LIB), Cint,
                (UInt64, UInt64, UInt64, UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint,
                 Cdouble, Cdouble, Cint, Cdouble, Cdouble, Ref{Cdouble}, Ref{Cdouble}),
                policy.behavior_actor.model.h, policy.behavior_critic.model.h, policy.target_actor.model.h,
                policy.target_critic.model.h, ds, da, dr, dt, dsn, Bu, policy.y, rho, 1,
                policy.behavior_actor.optimizer.eta, policy.behavior_critic.optimizer.eta, al, cl))
    foreach(device_free, (ds, da, dr, dt, dsn))
    policy.actor_loss = al[]
    policy.critic_loss = cl[]
end
# in Main, after including src/PDEagent.jl:
#   RLBase.update!(p::CustomDDPGPolicy{<:CustomNeuralNetworkApproximator{PDEenvHIP.HipMLP}}, batch::NamedTuple{SARTS}) =
#       PDEenvHIP.ddpg_update!(p, batch)

# ---- streams (the reference runs on one; a two-stream caller makes its streams here, BACK TO BACK: env, update, then the part
# streams an environment asks for -- hardware queues sit on the GPU's four compute pipes in the order they are made, see
# include/pdeconv.h at pdec_stream_create) ---------------------------------------------------------------------------------
function stream_create(level::Integer = 0)          # -1 high, 0 normal, +1 low; returns the hipStream_t
    s = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pdec_stream_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), s, level))
    s[]
end
stream_destroy(s::Ptr{Cvoid}) = check(ccall((:pdec_stream_destroy, LIB), Cint, (Ptr{Cvoid},), s))
function env_part_streams(env::UInt64)
    n = Ref{Cint}(0)
    check(ccall((:pdec_env_part_streams, LIB), Cint, (UInt64, Ref{Cint}), env, n))
    Int(n[])
end
env_set_part_streams(env::UInt64, streams::Vector{Ptr{Cvoid}}) =
    check(ccall((:pdec_env_set_part_streams, LIB), Cint, (UInt64, Ptr{Ptr{Cvoid}}, Cint), env, streams, length(streams)))

# ---- the reference-shaped training loop (one trajectory, `update_loops` x `batch_size` updates per step) with the trajectory on
# the device: what src/PDEagent.jl:237-361 does on host arrays, as launches.  A `DeviceTrajectory` holds the four circular traces
# (fp32, as RL.jl keeps them) and the two entry counters; RL.jl's stages map to
#   PRE_ACT   (:254-274)  push_sa!        POST_ACT (:276-289)  push_rt!       update!(policy) (:342-361)  update_small!
# and, where the library serves it (`step_glue!` returns true), the POST_ACT push of step t - 1, `agent(env)` and the PRE_ACT push
# of step t are ONE launch.  With the environment and the agent on two streams that run side by side (`streams_run_side_by_side`)
# the two hand-overs of a step can happen inside the kernels (`set_launch_sync!` on the actor's and the environment's handle
# before the glue launch / the env step): see run.py `_run_device_episodes` for the protocol, incl. the episode halt flag
# (`set_episode_halt!`) that lets a whole episode be enqueued without reading `is_terminated(env)` back per step.
mutable struct DeviceTrajectory
    h::UInt64                       # any handle on the stream the pushes run on (the actor's)
    state::Ptr{Cvoid}; action::Ptr{Cvoid}; reward::Ptr{Cvoid}; terminal::Ptr{Cvoid}
    capacity::Int64; stride::Int64; ns::Cint; na::Cint
    n_sa::Int64; n_rt::Int64
end
function push_sa!(t::DeviceTrajectory, s_dev::Ptr{Cvoid}, a_dev::Ptr{Cvoid}, n::Integer, dtype::Cint)
    cap1 = t.capacity + t.stride
    check(ccall((:pdec_replay_push_sa, LIB), Cint,
                (UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint),
                t.h, t.state, t.action, cap1, t.ns, t.na, t.n_sa % cap1, s_dev, a_dev, n, dtype))
    t.n_sa += n
end
function push_rt!(t::DeviceTrajectory, r_dev::Ptr{Cvoid}, done_dev::Ptr{Int32}, cols_per_traj::Integer, timeout::Bool, n::Integer, dtype::Cint)
    check(ccall((:pdec_replay_push_rt, LIB), Cint,
                (UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Int32}, Cint, Cint, Int64, Cint),
                t.h, t.reward, t.terminal, t.capacity, t.n_rt % t.capacity, r_dev, done_dev, cols_per_traj, timeout ? 1 : 0, n, dtype))
    t.n_rt += n
end
# update!(policy, trajectory) for batch_size <= 16: all `loops` minibatch updates in one launch, slots drawn on the device from
# the Philox stream (seed, offset); returns the offset for the next call
function update_small!(policy, t::DeviceTrajectory, loops::Integer, Bu::Integer, seed::UInt64, offset::UInt64, losses_dev::Ptr{Cvoid})
    rho = FROZEN_TARGETS[] ? 1.0 : Float64(policy.p)
    check(ccall((:pdec_ddpg_update_small_rng, LIB), Cint,
                (UInt64, UInt64, UInt64, UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, UInt64, UInt64,
                 Int64, Int64, Int64, Cint, Cdouble, Cdouble, Cint, Cdouble, Cdouble, Ptr{Cvoid}),
                policy.behavior_actor.model.h, policy.behavior_critic.model.h, policy.target_actor.model.h,
                policy.target_critic.model.h, t.state, t.action, t.reward, t.terminal, loops, Bu, seed, offset,
                min(t.n_rt, t.capacity), t.n_rt, t.capacity, t.stride, policy.y, rho, 1,
                policy.behavior_actor.optimizer.eta, policy.behavior_critic.optimizer.eta, losses_dev))
    offset + UInt64((loops * Bu + 3) ÷ 4)
end
# POST_ACT push of the step that ran (r_dev / done_dev, n_rt values; n_rt = 0: none) + agent(env) on state_dev [cols][ns]
# (act_mode 1: the actor with exploration noise, 2: the start policy's zero action) + PRE_ACT push of (state, action): one launch.
# false: not served (nothing enqueued, no counter moved): make the three calls.
function step_glue!(actor::HipMLP, t::DeviceTrajectory, dtype::Cint, r_dev::Ptr{Cvoid}, done_dev::Ptr{Int32}, cols_per_traj::Integer,
                    n_rt::Integer, act_mode::Integer, state_dev::Ptr{Cvoid}, cols::Integer, act_noise::Real, act_limit::Real,
                    seed::UInt64, offset::UInt64, actions_out::Ptr{Cvoid}, done_event::UInt64 = UInt64(0))
    served = Ref{Cint}(0)
    cap1 = t.capacity + t.stride
    check(ccall((:pdec_step_glue, LIB), Cint,
                (UInt64, UInt64, Cint, Ptr{Cvoid}, Ptr{Int32}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Cint,
                 Ptr{Cvoid}, Cint, Cdouble, Cdouble, UInt64, UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64,
                 UInt64, Ref{Cint}),
                actor.h, t.h, dtype, r_dev, done_dev, cols_per_traj, 0, t.reward, t.terminal, t.capacity, t.n_rt % t.capacity, n_rt,
                act_mode, state_dev, cols, act_noise, act_limit, seed, offset, actions_out, t.state, t.action, cap1,
                t.n_sa % cap1, cols, done_event, served))
    served[] == 1 || return false
    t.n_rt += n_rt
    t.n_sa += cols
    true
end
set_episode_halt!(h::UInt64, flag_dev::Ptr{Int32}) = check(ccall((:pdec_set_episode_halt, LIB), Cint, (UInt64, Ptr{Int32}), h, flag_dev))
set_launch_sync!(h::UInt64, wait_flag::Ptr{Int64}, wait_value::Integer, done_flag::Ptr{Int64}, done_value::Integer) =
    check(ccall((:pdec_set_launch_sync, LIB), Cint, (UInt64, Ptr{Int64}, Int64, Ptr{Int64}, Int64), h, wait_flag, wait_value, done_flag, done_value))
function streams_run_side_by_side(a::Ptr{Cvoid}, b::Ptr{Cvoid})
    yes = Ref{Cint}(0)
    check(ccall((:pdec_streams_run_side_by_side, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cint}), a, b, yes))
    yes[] == 1
end
function launch_sync_timeouts()
    n = Ref{Cint}(0)
    check(ccall((:pdec_launch_sync_timeouts, LIB), Cint, (Ref{Cint},), n))
    Int(n[])
end

# ---- multi-GPU (one Julia process per GPU; the reference itself is single-process) -----------------------------------
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:pdec_comm_unique_id, LIB), Cint, (Ptr{Cvoid},), id))
    id
end
function comm_create(nranks, rank, id::Vector{UInt8})
    c = Ref{UInt64}(0)
    check(ccall((:pdec_comm_create, LIB), Cint, (Ref{UInt64}, Cint, Cint, Ptr{Cvoid}), c, nranks, rank, id))
    c[]
end
# bounded wait for the rendezvous (a rank that never arrives must not park the others for ever)
function comm_create(nranks, rank, id::Vector{UInt8}, timeout_ms::Integer)
    c = Ref{UInt64}(0)
    check(ccall((:pdec_comm_create_timeout, LIB), Cint, (Ref{UInt64}, Cint, Cint, Ptr{Cvoid}, Cint), c, nranks, rank, id, timeout_ms))
    c[]
end
allreduce_grads(comm::UInt64, m::HipMLP) = check(ccall((:pdec_allreduce_grads, LIB), Cint, (UInt64, UInt64), comm, m.h))

end # module
