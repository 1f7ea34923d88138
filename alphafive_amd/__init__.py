"""alphafive_amd — MI355X-native self-play engine for GuoYi0/alphaFive's MCTS hot path.

Host-side mirror of the reference call surface (same names, arguments and error
behaviour) on top of the C-ABI HIP engine (include/af_engine.h):

    alphafive_amd.player.Player          genData/player.py:23   Player
    alphafive_amd.networkAPI.NetworkAPI  genData/networkAPI.py:10
    alphafive_amd.network.ResNet         genData/network.py:10  (PyTorch-ROCm + HIP)
    alphafive_amd.utils                  utils.py helpers + RandomStack record format
    alphafive_amd.engine.SelfPlayEngine  main.py:82 gen_data, batched on device

There is no CPU fallback: importing the engine without the built HIP library raises.
"""
__version__ = "0.1.0"
