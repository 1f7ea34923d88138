"""Default configuration — same attribute names and values as the reference's config.py:2-27,
so `import alphafive_amd.config as config` can be passed wherever the reference passes its
config module (Player(cfg=config, ...)).  Any attribute bag with these names works."""
board_size = 11
buffer_size = 12000
simulation_per_step = 542
upper_simulation_per_step = 642
goal = 5
batch_size = 512
lr_ = [(7000, 1e-3), (14000, 2e-4), (28000, 4e-5), (100000000, 2e-6)]
ckpt_path = "ckpt"
total_step = 20000
tau_decay_rate = 0.94
tau_decay_rate_r = 0.9
c_puct = 5.0
dirichlet_alpha = 0.3
gamma = 0.94
init_temp = 1.2
max_processes = 5


def get_lr(step):
    for bound, lr in lr_:
        if step < bound:
            return lr
    return lr_[-1][-1]
