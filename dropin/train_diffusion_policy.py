from ditreeonlineplanner_amd.train_diffusion_policy import init_noise_pred_net  # noqa: F401
