"""``init_noise_pred_net`` with the reference's signature (train_diffusion_policy.py:32-67); returns
the engine-backed parameter container instead of torch.nn layers.  Training is out of scope."""
from .model import NoisePredNet


def init_noise_pred_net(input_dim, action_dim, obs_dim, obs_history, action_history=0, goal_conditioned=True,
                        goal_dim=2, local_map_conditioned=True, local_map_encoder="identity",
                        local_map_embedding_dim=9, local_map_size=None, **kwargs):
    global_cond_dim = obs_dim * obs_history + goal_dim * goal_conditioned + action_history * action_dim
    if not local_map_conditioned or local_map_encoder.lower() != "resnet":
        raise NotImplementedError("the engine implements the 'resnet' local-map encoder (the reference default)")
    down_dims = tuple(kwargs.get("down_dims", (512, 1024, 2048)))
    return NoisePredNet(input_dim=input_dim, embedding_dim=local_map_embedding_dim,
                        additional_global_cond_dim=global_cond_dim, down_dims=down_dims,
                        local_map_size=local_map_size or 20)
