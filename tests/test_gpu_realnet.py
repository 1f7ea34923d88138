"""Config-1 path on the GPU box: Player(training=False, pv_fn=net.eval) with the shipped alphaFive-6960
weights through the HIP net, driven like self_play.py:79-106 — and the C oracle fed by the SAME
evaluator (one leaf at a time), so visit counts and moves must agree bit for bit even with the
real network."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, make_cfg

pytestmark = pytest.mark.gpu
W = os.path.join(GOLDEN, "alphaFive-6960.weights.npz")


def _hip_eval():
    import torch
    from alphafive_amd.network import ResNet
    net = ResNet(11, device="cuda")
    net.load_npz(W)
    pv = net.select_backend("hip")

    def eval_np(x):                      # numpy float32[B,3,S,S] -> (prob, value) numpy, like ResNet.eval
        p, v = pv(torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda())
        return p.cpu().numpy().copy(), v.cpu().numpy().copy()
    return net, eval_np


@pytest.mark.parametrize("training", [False, True])
def test_player_with_real_net_matches_oracle_through_same_evaluator(training):
    from alphafive_amd.player import Player
    from alphafive_amd import utils
    net, eval_np = _hip_eval()
    cfg = make_cfg(simulation_per_step=120, upper_simulation_per_step=160)
    pl = Player(cfg, training=training, pv_fn=eval_np, seed=3, game_id=1)
    orc = oracle.OraclePlayer(cfg, training=training, rng_mode=oracle.RNG_PHILOX, seed=3, game_id=1, pv_fn=eval_np)
    state, over, ply = pl.get_init_state(), False, 0
    last = None
    while not over and ply < 8:
        la = last if training else None           # self_play.py:95-97 passes None as last_action
        pol, act = pl.get_action(state, last_action=la)
        opol, oact, ovis = orc.get_action(state, la)
        assert (pl.last_visits == ovis).all(), f"ply {ply}: visit counts differ"
        assert act == oact
        if training:
            assert (pol.view(np.uint32) == opol.view(np.uint32)).all()
        else:
            assert pol is None and opol is None
        board = utils.step(utils.state_to_board(state, 11), act)
        state = utils.board_to_state(board)
        over, _ = utils.is_game_over(board, 5)
        last, ply = act, ply + 1
    assert len(pl.tree) == orc.tree_size()
    pl.close()


def _drive(pl, orc, training, plies, S=11, goal=5, state=None, last=None, on_ply=None):
    from alphafive_amd import utils
    state = pl.get_init_state() if state is None else state
    over, ply = False, 0
    while not over and ply < plies:
        la = last if training else None           # self_play.py:95-97 passes None as last_action
        pol, act = pl.get_action(state, last_action=la)
        opol, oact, ovis = orc.get_action(state, la)
        assert (pl.last_visits == ovis).all(), f"ply {ply}: visit counts differ"
        assert act == oact, f"ply {ply}"
        if training:
            assert (pol.view(np.uint32) == opol.view(np.uint32)).all(), f"ply {ply}: policy bits differ"
        else:
            assert pol is None and opol is None
        board = utils.step(utils.state_to_board(state, S), act)
        state = utils.board_to_state(board)
        over, _ = utils.is_game_over(board, goal)
        last, ply = act, ply + 1
        if on_ply is not None:
            on_ply(ply)
    return state, last


@pytest.mark.parametrize("training", [False, True])
def test_player_device_graph_path_matches_oracle_at_the_metric_settings(training):
    """The path that produces the config-1 number (self_play.py:79-106, player.py:128-147): Player(pv_fn=net.eval) with
    net on cuda keeps the leaves on the device, evaluates them with af_conv_f16s and replays 16 x (tick + forward) as a HIP
    graph — at BASELINE's own search settings (11x11, 500 simulations, cap 642, alphaFive-6960), against the C oracle fed
    leaf by leaf through a second handle of the same kernels (batch 1 on both sides).  Visit counts, moves, policy bits
    and the whole tree must agree bit for bit, and the graph must really be what ran."""
    from alphafive_amd.network import ResNet
    from alphafive_amd.player import Player
    from test_gpu_parity import _compare_tree
    import torch
    net = ResNet(11, device="cuda")
    net.load_npz(W)
    opv = net.select_backend("hip")               # the oracle's evaluator: its own handle of the same kernels

    def eval_np(x):
        p, v = opv(torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda())
        return p.cpu().numpy().copy(), v.cpu().numpy().copy()
    cfg = make_cfg(simulation_per_step=500, upper_simulation_per_step=642)
    pl = Player(cfg, training=training, pv_fn=net.eval, seed=11, game_id=2)
    assert pl.backend == "hip"
    orc = oracle.OraclePlayer(cfg, training=training, rng_mode=oracle.RNG_PHILOX, seed=11, game_id=2, pv_fn=eval_np)
    _drive(pl, orc, training, 8)
    assert pl._graph is not None and pl._graph[1] is not None and pl._graph[0][0][0] == int(training)      # the HIP-graph path ran
    _compare_tree(pl._engine.tree_dump(0), orc, 11)
    assert len(pl.tree) == orc.tree_size()
    pl.close()


def test_player_device_graph_path_on_15x15_matches_oracle():
    """BASELINE configs[3]'s board: the same device / HIP-graph Player path on 15x15 (4-word bitboards; the net's split-operand
    kernels run two half-board pseudo-positions per board), random-init weights, against the oracle fed by a second handle."""
    from alphafive_amd.network import ResNet
    from alphafive_amd.player import Player
    from test_gpu_parity import _compare_tree
    import torch
    net = ResNet(15, device="cuda", seed=21)
    opv = net.select_backend("hip")

    def eval_np(x):
        p, v = opv(torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda())
        return p.cpu().numpy().copy(), v.cpu().numpy().copy()
    cfg = make_cfg(board_size=15, simulation_per_step=200, upper_simulation_per_step=260)
    pl = Player(cfg, training=True, pv_fn=net.eval, seed=4, game_id=9)
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=4, game_id=9, pv_fn=eval_np)
    _drive(pl, orc, True, 4, S=15)
    assert pl.backend == "hip" and pl._graph is not None and pl._graph[1] is not None
    _compare_tree(pl._engine.tree_dump(0), orc, 15)
    pl.close()


def test_player_graph_path_follows_a_weight_update():
    """choose_best_player.py:78-82 reloads the nets between matches while the Players live on: a replayed graph skips the
    Python wrapper that notices net.restore()/load_npz()/set_variables(), so the graph is keyed on the weight version.
    After an update mid-game the Player must still equal the oracle fed by an evaluator with the NEW weights (the tree keeps
    the priors of the old ones on both sides)."""
    from alphafive_amd.network import ResNet, random_variables
    from alphafive_amd.player import Player
    from test_gpu_parity import _compare_tree
    import torch
    net = ResNet(11, device="cuda")
    net.load_npz(W)
    opv = net.select_backend("hip")

    def eval_np(x):
        p, v = opv(torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda())
        return p.cpu().numpy().copy(), v.cpu().numpy().copy()
    cfg = make_cfg(simulation_per_step=150, upper_simulation_per_step=200)
    pl = Player(cfg, training=True, pv_fn=net.eval, seed=5, game_id=7)
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=5, game_id=7, pv_fn=eval_np)
    state, last = _drive(pl, orc, True, 3)
    g0 = pl._graph
    net.set_variables(random_variables(11, seed=4))             # a different weight set, same shapes
    state, last = _drive(pl, orc, True, 3, state=state, last=last)
    assert pl._graph[0] != g0[0] and pl._graph[1] is not g0[1]  # dropped and captured again
    net.load_npz(W)
    _drive(pl, orc, True, 2, state=state, last=last)
    _compare_tree(pl._engine.tree_dump(0), orc, 11)
    pl.close()


def test_player_graph_path_follows_the_live_simulation_budget():
    """player.py:140-143 reads config.simulation_per_step / upper_simulation_per_step at every get_action.  EngineParams travel by
    value, so a captured graph has the budget baked in: the graph is keyed on Engine.params_key() and captured again when the
    caller changes the budget between two moves (the stale graph kept replaying the old budget)."""
    from alphafive_amd.network import ResNet
    from alphafive_amd.player import Player
    from test_gpu_parity import _compare_tree
    import torch
    net = ResNet(11, device="cuda")
    net.load_npz(W)
    opv = net.select_backend("hip")

    def eval_np(x):
        p, v = opv(torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda())
        return p.cpu().numpy().copy(), v.cpu().numpy().copy()
    cfg = make_cfg(simulation_per_step=120, upper_simulation_per_step=160)
    pl = Player(cfg, training=True, pv_fn=net.eval, seed=8, game_id=1)
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=8, game_id=1, pv_fn=eval_np)
    state, last, keys = None, None, []
    for sims, upper in ((120, 160), (30, 40), (200, 420), (120, 160)):
        cfg.simulation_per_step, cfg.upper_simulation_per_step = sims, upper
        orc.set_simulations(sims, upper)
        state, last = _drive(pl, orc, True, 2, state=state, last=last)
        assert pl._graph is not None and pl._graph[0][0][1:3] == (sims, upper)
        keys.append(pl._graph[0])
    assert len(set(keys)) == 3
    _compare_tree(pl._engine.tree_dump(0), orc, 11)
    pl.close()


def test_resnet_eval_and_networkapi_run_the_hand_written_kernels():
    """networkAPI.py:64-75: the worker thread calls agent_model.eval(batch).  On cuda that is the af_conv_f16s path — the
    same bits as select_backend("hip") on the same batch — not vendor ops; eval_torch() is the explicit reference."""
    from alphafive_amd.network import ResNet
    from alphafive_amd.networkAPI import NetworkAPI
    import torch
    net = ResNet(11, device="cuda")
    net.load_npz(W)
    x = _rand_planes(5, 11, seed=3)
    p, v = net.eval(x)
    ph, vh = net.select_backend("hip")(torch.from_numpy(x).cuda())
    assert (p == ph.cpu().numpy()).all() and (v == vh.cpu().numpy()).all()
    pt, vt = net.eval_torch(torch.from_numpy(x).cuda())
    assert np.abs(p - pt.cpu().numpy()).max() < 5e-5 and not (p == pt.cpu().numpy()).all()     # a different arithmetic
    cfg = make_cfg(max_processes=3)
    api = NetworkAPI(cfg, net)
    api.start(reload=False)
    pipes = [api.get_pipe(reload=False) for _ in range(3)]
    for i, pp in enumerate(pipes):
        pp.send([x[i]])
    for i, pp in enumerate(pipes):
        assert pp.poll(20)
        pol, val = pp.recv()[0]
        p1, v1 = net.eval(x[i:i + 1])                           # batch-slot independent: same bits alone
        assert (np.asarray(pol, np.float32) == p1[0]).all() and val == float(v1[0])
    api.close()


def test_headless_self_play_loop_finishes_a_game():
    from alphafive_amd import self_play
    from alphafive_amd.network import ResNet
    net = ResNet(11, device="cuda")
    net.load_npz(W)
    cfg = make_cfg(simulation_per_step=60, upper_simulation_per_step=80)
    moves, value, dt = self_play.play(cfg, net, seed=0, verbose=False)
    assert 9 <= len(moves) <= 121 and value in (-1.0, 0.0)
    assert len(set(moves)) == len(moves)          # no cell played twice


def test_selfplay_engine_with_real_net_produces_valid_episodes():
    import torch
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet
    from alphafive_amd import utils
    net = ResNet(11, device="cuda")
    net.load_npz(W)
    cfg = make_cfg(simulation_per_step=40, upper_simulation_per_step=60)
    sp = SelfPlayEngine(cfg, 256, net.select_backend("hip"), device=0, seed=9)
    eps = []
    for _ in range(60):
        sp.run_ticks(100)
        sp.check()
        eps += sp.pop_episodes(256)
        if len(eps) >= 64:
            break
    assert len(eps) >= 64
    for rec, result in eps[:64]:
        T = len(rec)
        assert 9 <= T <= 121 and result in (-1, 0, 1)
        board = np.zeros((11, 11), np.int8)
        for t, (s, p, la, v, w) in enumerate(rec):
            assert s == utils.board_to_state(board)          # states chain by the recorded moves
            assert p.shape == (11, 11) and abs(float(p.sum()) - 1.0) < 1e-4 and isinstance(w, np.float32)
            assert (p[board != 0] == 0).all()
            if t + 1 < T:
                nxt = rec[t + 1][2]
                board = utils.step(board, nxt)
        if result != 0:
            assert rec[-1][3] == 1.0 and (result == 1) == (T % 2 == 1)     # last mover won (player.py:74-76)
    ct = sp.counters()
    assert ct["terminals"] > 0 and ct["episodes"] >= 64
    sp.close()


def test_selfplay_engine_on_15x15_with_the_split_operand_net():
    """BASELINE configs[3] shape in small: the batched engine on 15x15 boards (4-word bitboards) with the hand-written net
    (two half-board pseudo-positions per board, fused heads) behind the evaluator seam: the leaves the engine parks are evaluated
    within 1e-5 of the fp64 restatement, and complete episodes chain correctly."""
    import torch
    from oracle import net_fp64
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet
    from alphafive_amd import utils
    S, G = 15, 192
    net = ResNet(S, device="cuda", seed=15)
    pv = net.select_backend("hip")
    cfg = make_cfg(board_size=S, simulation_per_step=24, upper_simulation_per_step=32)
    sp = SelfPlayEngine(cfg, G, pv, device=0, seed=3)
    sp.run_ticks(80)
    sp.check()
    x = sp.planes[:6].cpu().numpy()
    p, v = pv(sp.planes)
    p64, v64 = net_fp64.forward(net.variables, x)
    assert np.abs(v[:6].cpu().numpy() - v64).max() < 1e-5 and np.abs(p[:6].cpu().numpy() - p64).max() < 1e-5
    eps = []
    for _ in range(80):
        sp.run_ticks(200)
        sp.check()
        eps += sp.pop_episodes(256)
        if len(eps) >= 24:
            break
    assert len(eps) >= 24
    for rec, result in eps[:24]:
        T = len(rec)
        assert 9 <= T <= S * S and result in (-1, 0, 1)
        board = np.zeros((S, S), np.int8)
        for t, (s_, pol, la, val, w) in enumerate(rec):
            assert s_ == utils.board_to_state(board) and pol.shape == (S, S) and abs(float(pol.sum()) - 1.0) < 1e-4
            assert (pol[board != 0] == 0).all()
            if t + 1 < T:
                board = utils.step(board, rec[t + 1][2])
    sp.close()


def _rand_planes(B, S, seed):
    rng = np.random.RandomState(seed)
    x = np.zeros((B, 3, S, S), np.float32)
    for b in range(B):
        n = rng.randint(0, S * S - 1)
        cells = rng.permutation(S * S)[:n + 1]
        x[b, 0].reshape(-1)[cells[0:n:2]] = 1
        x[b, 1].reshape(-1)[cells[1:n:2]] = 1
        x[b, 2].reshape(-1)[cells[n]] = 1
    return x


@pytest.mark.parametrize("blocks", [3, 8])
def test_deep_bf16_tower_kernel_matches_fp32_reference(blocks):
    """af_tower_forward (hand-written bf16 MFMA tower, BASELINE configs[4]) against an fp32 PyTorch evaluation of the
    same bf16 weights with the activations rounded to bf16 at the same points (after each ELU).  Bar: the kernel's error
    must not exceed the error of the all-PyTorch bf16 path it replaces (bf16 keeps 8 mantissa bits; no bit parity)."""
    import torch
    import torch.nn.functional as F
    from alphafive_amd.network_deep import DeepResNet
    B = 300                                          # more than one pass of the 256 persistent workgroups, ragged tail
    net = DeepResNet(11, blocks=blocks, width=128, device="cuda", seed=3)      # 8 = BASELINE configs[4]'s depth
    g = torch.Generator().manual_seed(5)
    for blk in net.tower:                            # non-zero biases
        for k in ("res", "c1", "c2"):
            blk[k] = (blk[k][0], (torch.randn(128, generator=g) * 0.1).to("cuda", torch.bfloat16))
    pv = net.select_backend("hip", B)
    tw = net._tower
    h0 = (torch.randn((B, 128, 11, 11), generator=g) * 0.5).to("cuda", torch.bfloat16)
    ref = h0.float()
    for blk in net.tower:
        w = {k: (blk[k][0].float(), blk[k][1].float()) for k in blk}
        mid = F.elu(F.conv2d(ref, *w["c1"], padding=1)).bfloat16().float()
        ref = F.elu(F.conv2d(ref, *w["res"]) + F.conv2d(mid, *w["c2"], padding=1)).bfloat16().float()
    tw.load_nchw(h0)
    tw.forward(B)
    out = tw.store_nchw(B).float()
    torch_bf16 = net.tower_reference(h0).float()
    err_hip, err_torch = (out - ref).abs(), (torch_bf16 - ref).abs()
    assert torch.isfinite(out).all()
    assert err_hip.mean().item() <= 1.1 * err_torch.mean().item() + 1e-4
    assert err_hip.max().item() <= 2.0 * err_torch.max().item() + 0.02
    S = 11                                           # the zero rows above / below the board are never written
    for buf in (tw.x, tw.g):
        assert float(buf[:, :, :S].float().abs().sum()) == 0.0 and float(buf[:, :, S + S * S:].float().abs().sum()) == 0.0
    tw.load_nchw(h0)                                 # a second pass reproduces the result bit for bit
    tw.forward(B)
    assert torch.equal(tw.store_nchw(B).float(), out)
    from alphafive_amd import tower_hip              # the other convolution kernels (af_tower_tune key 3) meet the same bar
    for engine in (0, 2):                            # 0 af_tower_conv for both convolutions, 2 af_tower_conv3 for both
        try:
            tower_hip.tune(3, engine)
            tw.load_nchw(h0)
            tw.forward(B)
            out2 = tw.store_nchw(B).float()
        finally:
            tower_hip.tune(3, 3)
        err2 = (out2 - ref).abs()
        assert err2.mean().item() <= 1.1 * err_torch.mean().item() + 1e-4 and err2.max().item() <= 2.0 * err_torch.max().item() + 0.02, engine
        if engine == 0:                              # the default (conv3 for a block's first convolution only) changes no bit against it
            assert torch.equal(out2, out)
        for buf in (tw.x, tw.g):
            assert float(buf[:, :, :S].float().abs().sum()) == 0.0 and float(buf[:, :, S + S * S:].float().abs().sum()) == 0.0
    x = torch.from_numpy(_rand_planes(B, 11, seed=2)).cuda()
    # stem kernel and heads' 1x1-conv kernel vs fp32 PyTorch ops on the same (bf16-valued) weights and inputs
    net.stem = (net.stem[0], (torch.randn(128, generator=g) * 0.1).to("cuda", torch.bfloat16))
    net.vconv = (net.vconv[0], (torch.randn(4, generator=g) * 0.1).to("cuda", torch.bfloat16))
    net.pconv = (net.pconv[0], (torch.randn(16, generator=g) * 0.1).to("cuda", torch.bfloat16))
    pv = net.select_backend("hip", B)
    tw = net._tower
    tw.stem(x)
    stem_ref = F.elu(F.conv2d(x, net.stem[0].float(), net.stem[1].float(), padding=2))
    assert (tw.store_nchw(B).float() - stem_ref).abs().max().item() <= 2.0 ** -8 * stem_ref.abs().max().item() + 1e-6
    vin, pin = tw.heads(B)
    h = tw.store_nchw(B).float()
    v_ref = F.elu(F.conv2d(h, net.vconv[0].float(), net.vconv[1].float())).reshape(B, -1)
    p_ref = F.elu(F.conv2d(h, net.pconv[0].float(), net.pconv[1].float())).reshape(B, -1)
    assert (vin.float() - v_ref).abs().max().item() <= 2.0 ** -8 * v_ref.abs().max().item() + 1e-6
    assert (pin.float() - p_ref).abs().max().item() <= 2.0 ** -8 * p_ref.abs().max().item() + 1e-6
    try:                                             # (af_tower_tune key 4: 0 = the VALU heads kernel the MFMA one replaced in r4)
        tower_hip.tune(4, 0)
        vin0, pin0 = (t.clone() for t in tw.heads(B))
    finally:
        tower_hip.tune(4, 1)
    assert (vin0.float() - v_ref).abs().max().item() <= 2.0 ** -8 * v_ref.abs().max().item() + 1e-6
    assert (pin0.float() - p_ref).abs().max().item() <= 2.0 ** -8 * p_ref.abs().max().item() + 1e-6
    # the MFMA dense kernel (fc1+ELU, fc2+tanh(x/2), policy fc + softmax) vs the same layers on PyTorch ops: bf16 inputs and
    # weights on both sides; the kernel keeps fp32 between the layers where the torch path rounds to bf16
    net.vfc1 = (net.vfc1[0], (torch.randn(64, generator=g) * 0.1).to("cuda", torch.bfloat16))
    net.pfc = (net.pfc[0], (torch.randn(121, generator=g) * 0.1).to("cuda", torch.bfloat16))
    pv = net.select_backend("hip", B)
    pd, vd = (t.clone() for t in pv(x))
    pt, vt = net.eval_hip_torch_dense(x)
    assert pd.shape == (B, 121) and (pd.sum(1) - 1).abs().max().item() < 1e-5
    assert (pd - pt).abs().max().item() < 2e-3 and (vd - vt).abs().max().item() < 2e-2
    assert pd.argmax(1).eq(pt.argmax(1)).float().mean().item() > 0.98
    for Bs in (1, 31, 33):                                       # ragged tails of the 32-position workgroups
        ps, vs = pv(x[:Bs].contiguous())
        assert torch.equal(ps, pd[:Bs]) and torch.equal(vs, vd[:Bs])
    p, v = (t.clone() for t in pv(x))                            # end to end through the evaluator seam
    assert p.shape == (B, 121) and torch.allclose(p.sum(1), torch.ones(B, device="cuda"), atol=1e-3)
    _assert_no_worse_than_torch_bf16(net, x, p, v)


def _assert_no_worse_than_torch_bf16(net, x, p, v):
    """Config 5 has no bit parity (bf16, random init); its bar: per output, the hand-written path's error against
    an fp32 evaluation of the same bf16-valued weights is bounded by the error of the all-PyTorch bf16 path it replaces on the very
    same positions — mean <= 1.25x, max <= 2x (+ one bf16 ulp of the output range) — instead of a fixed 0.05 / 0.1."""
    import torch
    p32, v32 = net.eval_device(x, dtype=torch.float32)
    pt, vt = net.eval_device(x)
    for name, got, ref, tor, ulp in (("policy", p, p32, pt, 2.0 ** -9 * float(p32.max())), ("value", v, v32, vt, 2.0 ** -9)):
        e_hip, e_tor = (got.float() - ref).abs(), (tor.float() - ref).abs()
        print("config 5 %s: |hip - fp32| mean %.3g max %.3g; |torch bf16 - fp32| mean %.3g max %.3g"
              % (name, e_hip.mean().item(), e_hip.max().item(), e_tor.mean().item(), e_tor.max().item()))
        assert e_hip.mean().item() <= 1.25 * e_tor.mean().item() + 1e-7, name
        assert e_hip.max().item() <= 2.0 * e_tor.max().item() + ulp, name
    assert p.argmax(1).eq(p32.argmax(1)).float().mean().item() >= pt.argmax(1).eq(p32.argmax(1)).float().mean().item() - 0.02


@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_deep_bf16_evaluator_plugs_into_the_engine(backend):
    """BASELINE configs[4] shape: the 8-block width-128 bf16 net (hand-written tower, or all PyTorch ops) behind the
    same evaluator seam as the fp32 net (performance-only config)."""
    import torch
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network_deep import DeepResNet
    net = DeepResNet(11, blocks=8, width=128, device="cuda")
    pv = net.select_backend(backend, 512)
    cfg = make_cfg(simulation_per_step=30, upper_simulation_per_step=40)
    sp = SelfPlayEngine(cfg, 512, pv, device=0, seed=2)
    sp.run_ticks(40)
    sp.check()
    p, v = pv(sp.planes)
    assert p.dtype == torch.float32 and p.shape == (512, 121) and torch.isfinite(p).all() and torch.isfinite(v).all()
    assert (p.sum(1) - 1).abs().max().item() < 1e-3 and v.abs().max().item() <= 1.0
    ct = sp.counters()
    assert ct["sims"] == 512 * 39 and ct["plies"] == 512          # 30 sims -> one move each, then 10 more sims
    sp.close()


def test_player_pipe_mode_through_networkapi_matches_the_pipe_oracle():
    """player.py:194-197 pipe protocol end to end: Player(pipe=net.get_pipes(cfg)) against the NetworkAPI worker thread +
    ResNet.eval.  Behind a pipe the values are python floats (networkAPI.py:72), so the reference's W / Q are fp64 there
    (SURVEY 8a rule 2): the engine switches to its fp64 store and must equal the oracle's pipe variant — itself pinned on
    the reference Player behind the reference NetworkAPI (tests/golden/*_pipe.npz) — fed by the same ResNet.eval."""
    import oracle
    from alphafive_amd.network import ResNet
    from alphafive_amd.player import Player
    from alphafive_amd import utils
    S = 6                                    # (a 6x6 net of the same architecture, random init: every evaluation crosses the
    net = ResNet(S, device="cuda", seed=3)   # pipe and two GIL hand-overs, so the test is sized by its evaluation count)
    cfg = make_cfg(board_size=S, goal=4, simulation_per_step=100, upper_simulation_per_step=150)   # > 2 x 36: past the forced root visits
    pipe = net.get_pipes(cfg)
    a = Player(cfg, training=True, pipe=pipe, seed=4, game_id=0)
    assert a._engine.value_f64
    orc = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=4, game_id=0, pv_fn=net.eval, value_f64=True)
    state, last = a.get_init_state(), None
    for _ in range(3):
        pa, aa = a.get_action(state, last_action=last)
        po, ao, vo = orc.get_action(state, last)
        assert aa == ao and (a.last_visits == vo).all()
        assert (pa.view(np.uint32) == po.view(np.uint32)).all()
        board = utils.step(utils.state_to_board(state, S), aa)
        state, last = utils.board_to_state(board), aa
    dd, od = a._engine.tree_dump(0), orc.tree_dump()
    omap = {od["keys"][i][[0, 1, 4, 5]].tobytes(): i for i in range(len(od["sum_n"]))}     # 2-word keys (36 cells)
    rounded = 0
    for i in range(len(dd["sum_n"])):
        j = omap[dd["keys"][i].tobytes()]
        assert (dd["n"][i] == od["n"][j]).all() and not dd["f32"][i].any()
        assert dd["w"].dtype == np.float64 and (dd["w"][i] == od["w64"][j]).all()
        rounded += int((dd["w"][i] != dd["w"][i].astype(np.float32)).sum())
    assert rounded > 0                       # the sums really left the fp32 grid: an fp32 store could not have matched
    a.close()
    net.close()


def test_config5_full_size_8192_games_on_the_bf16_tower():
    """BASELINE configs[4] at its own size: 8192 concurrent games on the hand-written bf16 tower for more than one full
    ply (performance-only configuration: no bit parity for the net, but the tree side is the same bit-exact engine, so the
    bookkeeping invariants must hold and the evaluator must stay a probability distribution on all 8192 leaves)."""
    import torch
    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network_deep import DeepResNet
    G = 8192
    net = DeepResNet(11, blocks=8, width=128, device="cuda")
    pv = net.select_backend("hip", G)
    cfg = make_cfg(simulation_per_step=60, upper_simulation_per_step=80)
    sp = SelfPlayEngine(cfg, G, pv, device=0, seed=6)
    sp.run_ticks(150)                                 # 60 sims -> first move, 60 -> second, 30 into the third
    sp.check()
    ct = sp.counters()
    assert ct["plies"] == 2 * G and ct["sims"] == ct["expands"] + ct["terminals"] and ct["sims"] >= 148 * G
    p, v = pv(sp.planes)
    assert p.shape == (G, 121) and torch.isfinite(p).all() and torch.isfinite(v).all()
    assert (p.sum(1) - 1).abs().max().item() < 1e-3 and v.abs().max().item() <= 1.0
    # on the very leaves of this batch the hand-written path is no further from fp32 arithmetic than the all-PyTorch bf16 path
    _assert_no_worse_than_torch_bf16(net, sp.planes, p.clone(), v.clone())
    d = sp.engine.tree_dump(G - 1)
    assert ((d["sum_n"] - d["n"].sum(1) >= 0) & (d["sum_n"] - d["n"].sum(1) <= 1)).all()
    sp.close()
