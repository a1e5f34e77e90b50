# KSSetupHIP.jl -- the lines a maintainer changes in scripts/KS/setup/KSSetup.jl to run the KS experiments on
# libpdeconv.so.  Include it AFTER the setup's globals and prepare_gaussians (KSSetup.jl:20-113) are defined; it
# replaces the `do_step` closure of KSSetup.jl:130-160.  `PDEenv(do_step = do_step, ...)` (KSSetup.jl:251-262),
# `PDEhook` and `run(agent, env, stop_condition, hook)` stay as they are.
include(pwd() * "/src/PDEenvHIP.jl")
PDEenvHIP.init(0)

const HENV_CFG = PDEenvHIP.ks_cfg(nx = nx, Lx = Lx, dt = dt, oversampling = oversampling, mu = μ, max_value = max_value,
                                  agent_power = agent_power, action_punish = action_punish,
                                  delta_action_punish = delta_action_punish, n_sensors = length(sensor_positions),
                                  n_actuators = length(actuator_positions), window_size = window_size,
                                  temporal_steps = temporal_steps)
const HENV = PDEenvHIP.env_create(HENV_CFG, gaussians, gaussians_actuators, actuators_to_sensors)

do_step(env) = PDEenvHIP.do_step(HENV, env.y, env.p)          # was scripts/KS/setup/KSSetup.jl:130-160

# the networks: wrap the chains built by create_NNA (src/PDEagent.jl:46-49)
#   model = PDEenvHIP.HipMLP(create_chain(...); max_cols = batch_size)
# and route update!(policy, batch) to the fused call (see the end of PDEenvHIP.jl).
