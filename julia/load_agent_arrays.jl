# load_agent_arrays.jl -- read an agent / actor that the MI355X build saved with checkpoint.save_agent_jld2 /
# save_actor_jld2 (plain Float32 / Float64 / Int64 arrays in a JLD2 file) back into the reference's Julia objects.
# Not executed in the build image (no Julia there); the file layout is produced and re-read by
# distributedconvrl-pde-control_amd/jld2.py (tests/test_host_logic.py::test_jld2_writer_roundtrip_and_checksums).
using JLD2, Flux

const ACT = Dict(0 => identity, 1 => relu, 2 => tanh)

# Chain(Dense...) from `<prefix>_W1`, `<prefix>_b1`, ... (W is stored [out, in] exactly as Flux keeps Dense.weight)
function chain_from_arrays(d::Dict, prefix::String)
    acts = d["$(prefix)_acts"]
    layers = [Dense(d["$(prefix)_W$i"], d["$(prefix)_b$i"], ACT[Int(acts[i])]) for i in 1:length(acts)]
    Chain(layers...)
end

# copyto!(agent.policy.behavior_actor, ...) for the four networks of an Agent built by create_agent (src/PDEagent.jl:58-119)
function load_agent_arrays!(agent, path::String)
    d = load(path)
    for (name, nna) in (("behavior_actor", agent.policy.behavior_actor), ("behavior_critic", agent.policy.behavior_critic),
                        ("target_actor", agent.policy.target_actor), ("target_critic", agent.policy.target_critic))
        Flux.loadparams!(nna.model, Flux.params(chain_from_arrays(d, name)))
    end
    agent
end

# d["hyper"] = [gamma, rho, act_limit, act_noise, eta_actor, eta_critic, rho_effective]; rho_effective == 1.0: the run kept its
# target networks frozen (PDEenvHIP.FROZEN_TARGETS, the reference as its committed source runs), so the target_* arrays are the
# initial networks; otherwise they are Polyak averages with that factor
target_regime(path::String) = (h = load(path)["hyper"]; length(h) >= 7 && h[7] == 1.0 ? :frozen : :moving)

# hook.bestNNA for plot_heat (src/plotting.jl:26-31): copyto!(agent.policy.behavior_actor, best)
best_actor(path::String; name = "bestNNA") = chain_from_arrays(load(path), name)
