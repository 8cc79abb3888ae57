// get_agent_factory (evo_motion_networks/src/agent_factory.cpp:186-211) for the agents of this path: "ppo_gae", "soft_actor_critic",
// "random", "constant" (the reference lists ten; an unknown name -> std::invalid_argument, :208-209).  Parameter keys are the
// reference's (agent_factory.cpp:66-80,112-125,137-146); "device" (and "seed" of the random agent) are this adapter's own.
#pragma once
#include "ppo_gae_agent_hip.hpp"
#include "sac_agent_hip.hpp"

namespace evm_adapter {

inline std::shared_ptr<AgentFactoryHip> get_agent_factory(const std::string &agent_name, std::map<std::string, std::string> parameters) {
    if (agent_name == "ppo_gae") return std::make_shared<PpoGaeHipFactory>(std::move(parameters));
    if (agent_name == "soft_actor_critic") return std::make_shared<SoftActorCriticHipFactory>(std::move(parameters));
    if (agent_name == "random") return std::make_shared<RandomAgentHipFactory>(std::move(parameters));
    if (agent_name == "constant") return std::make_shared<ConstantAgentHipFactory>(std::move(parameters));
    throw std::invalid_argument(agent_name);
}

}  // namespace evm_adapter
