"""CPU: host-side mirror of the reference interface -- registry names,
constructor signatures and state_dict keys/shapes (pinned by the name/shape
lists captured from the reference's own classes in F_NET / F_GEN / F_DEC)."""
import os

import numpy as np
import pytest

from rfdnet_amd.iscnet.config import Config


def ref_keys(fx, prefix):
    return [(str(n), tuple(int(x) for x in str(s).split(",")) if str(s) else ())
            for n, s in zip(fx[prefix + "_names"], fx[prefix + "_shapes"])]


def my_keys(m):
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]


@pytest.fixture(scope="module")
def fnet(golden_dir):
    return np.load(os.path.join(golden_dir, "F_NET.npz"))


def test_registry_has_the_reference_names():
    from rfdnet_amd.iscnet import network  # noqa: F401
    from rfdnet_amd.iscnet.registers import METHODS, MODULES
    assert sorted(MODULES.module_dict) == ['ONet', 'Pointnet2Backbone', 'ProposalModule',
                                           'SkipPropagation', 'VotingModule']
    assert list(METHODS.module_dict) == ['ISCNet']
    assert MODULES.get('nope', 'ONet') is MODULES.get('ONet')       # alter_key behaviour, registry.py:27-31
    with pytest.raises(KeyError):
        MODULES.register_module(MODULES.get('ONet'))


def test_backbone_keys(fnet):
    from rfdnet_amd.iscnet.pointnet2backbone import Pointnet2Backbone
    assert my_keys(Pointnet2Backbone(Config())) == ref_keys(fnet, "bb")


def test_voting_keys(fnet):
    from rfdnet_amd.iscnet.vote_module import VotingModule
    assert my_keys(VotingModule(Config())) == ref_keys(fnet, "vote")


def test_proposal_keys(fnet):
    from rfdnet_amd.iscnet.proposal_module import ProposalModule
    assert my_keys(ProposalModule(Config())) == ref_keys(fnet, "prop")


def test_skip_propagation_keys(fnet):
    from rfdnet_amd.iscnet.skip_propagation import SkipPropagation
    assert my_keys(SkipPropagation(Config())) == ref_keys(fnet, "skip")


def test_onet_decoder_keys_are_the_reference_subset(golden_dir):
    """the reference ONet additionally owns encoder_latent.* (training-only, q(z|.));
    everything else must match key for key"""
    from rfdnet_amd.iscnet.occupancy_net import ONet
    fx = np.load(os.path.join(golden_dir, "F_GEN.npz"))
    ref = [kv for kv in ref_keys(fx, "onet") if not kv[0].startswith("encoder_latent.")]
    assert my_keys(ONet(Config())) == ref


def test_iscnet_assembles_by_phase_and_loads_reference_style_checkpoint():
    import torch
    from rfdnet_amd.iscnet.network import ISCNet
    net = ISCNet(Config())
    assert [n for n, _ in net.named_children()] == ['backbone', 'voting', 'detection',
                                                    'skip_propagation', 'completion']
    det = ISCNet(Config({'demo': {'phase': 'detection'}}))
    assert [n for n, _ in det.named_children()] == ['backbone', 'voting', 'detection']
    sd = {'module.' + k: torch.full_like(v, 2) for k, v in net.state_dict().items()
          if k.startswith('voting.')}
    sd['module.not_a_key'] = torch.zeros(1)
    net.load_weight(sd)                                   # models/network.py:81-89 semantics
    assert float(net.voting.conv1.weight.mean()) == 2.0


def test_query_and_group_rejects_unbuilt_options():
    from rfdnet_amd.pointnet2_ops.pointnet2_utils import QueryAndGroup
    with pytest.raises(NotImplementedError):
        QueryAndGroup(0.2, 16, sample_uniformly=True)


def test_generator_logit_threshold_and_unbuilt_options():
    from rfdnet_amd.iscnet.generator import Generator3D
    g = Generator3D(None, threshold=0.5, resolution0=32, upsampling_steps=1)
    assert g.logit_threshold() == 0.0                      # generator.py:85
    assert abs(Generator3D(None, threshold=0.2).logit_threshold() - np.log(0.25)) < 1e-12
    with pytest.raises(NotImplementedError):
        Generator3D(None, refinement_step=3)


def test_synthetic_scene_is_reproducible_and_has_the_awkward_points():
    from rfdnet_amd import synthetic
    a = synthetic.synthetic_scene(seed=10, n_raw=30000, n_points=40000)
    b = synthetic.synthetic_scene(seed=10, n_raw=30000, n_points=40000)
    assert np.array_equal(a, b) and a.shape == (40000, 4) and a.dtype == np.float32
    assert len(np.unique(a, axis=0)) < 40000               # sampled with replacement => duplicates
    assert ((a[:, :3] ** 2).sum(1) <= 1e-3).sum() >= 1     # near-origin points (FPS skip rule)


def test_mean_size_array_sources(tmp_path, monkeypatch):
    """scannet_config.py:21 reads datasets/scannet/scannet_means.npz relative to the working directory; here an
    explicit array / path wins, then $RFD_MEAN_SIZE_NPZ, then that relative path, else a flagged placeholder."""
    import numpy as np
    from rfdnet_amd.iscnet.config import Config, ScannetConfig
    monkeypatch.delenv("RFD_MEAN_SIZE_NPZ", raising=False)
    monkeypatch.chdir(tmp_path)
    assert ScannetConfig().placeholder_sizes
    arr = np.arange(24, dtype=np.float64).reshape(8, 3) / 10
    assert np.array_equal(ScannetConfig(arr).mean_size_arr, arr) and not ScannetConfig(arr).placeholder_sizes
    np.savez(tmp_path / "m.npz", arr)                       # key 'arr_0', like the reference's file
    assert np.array_equal(Config(mean_size_arr=str(tmp_path / "m.npz")).dataset_config.mean_size_arr, arr)
    monkeypatch.setenv("RFD_MEAN_SIZE_NPZ", str(tmp_path / "m.npz"))
    assert np.array_equal(ScannetConfig().mean_size_arr, arr)
    monkeypatch.delenv("RFD_MEAN_SIZE_NPZ")
    (tmp_path / "datasets" / "scannet").mkdir(parents=True)
    np.savez(tmp_path / "datasets" / "scannet" / "scannet_means.npz", arr + 1)
    c = ScannetConfig()
    assert np.array_equal(c.mean_size_arr, arr + 1) and not c.placeholder_sizes


REF_YAML = "/root/reference/configs/config_files/ISCNet_test.yaml"


def test_config_from_a_reference_style_yaml(tmp_path):
    """Config.from_yaml reads the reference's config FILE FORMAT (keys of configs/config_files/ISCNet_test.yaml),
    sets the mode like main.py and mounts the eval dictionary like config_utils.mount_external_config."""
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    y = tmp_path / "cfg.yaml"
    y.write_text("""
method: ISCNet
weight: ['out/pretrained_models/pretrained_weight.pth']
seed: 10
device: {use_gpu: True, gpu_ids: '0', num_workers: 0}
data: {dataset: scannet, num_point: 40000, num_target: 256, vote_factor: 1, cluster_sampling: seed_fps,
       no_height: False, use_color_detection: False, use_color_completion: False, hidden_dim: 512, c_dim: 512,
       z_dim: 32, threshold: 0.5, use_cls_for_completion: False, skip_propagate: True}
model:
  backbone: {method: Pointnet2Backbone, loss: Null}
  voting: {method: VotingModule, loss: Null}
  detection: {method: ProposalModule, loss: DetectionLoss}
  skip_propagation: {method: SkipPropagation, loss: Null}
  completion: {method: ONet, loss: ONet_Loss, weight: 0.005}
test: {phase: 'completion', batch_size: 1, use_cls_nms: False, use_3d_nms: True, faster_eval: True, nms_iou: 0.3,
       use_old_type_nms: False, per_class_proposal: True, conf_thresh: 0.07}
generation: {generate_mesh: True, resolution_0: 16, upsampling_steps: 2, use_sampling: False, refinement_step: 0,
             simplify_nfaces: Null, dump_threshold: 0.4, dump_results: True}
demo: {phase: 'completion'}
log: {vis_path: visualization, path: out/iscnet}
""")
    cfg = Config.from_yaml(str(y), mode="demo")
    assert cfg.config["mode"] == "demo" and cfg.config["data"]["num_point"] == 40000
    assert cfg.config["generation"]["resolution_0"] == 16 and cfg.config["generation"]["dump_threshold"] == 0.4
    assert cfg.eval_overrides == {"remove_empty_box": False, "use_3d_nms": True, "nms_iou": 0.3,
                                  "use_old_type_nms": False, "cls_nms": False, "per_class_proposal": True,
                                  "conf_thresh": 0.07}
    net = ISCNet(cfg)                                        # builds the five sub-networks by registry name
    assert [n for n, _ in net.named_children()] == ["backbone", "voting", "detection", "skip_propagation", "completion"]
    assert net.completion.generator.resolution0 == 16 and net.completion.generator.upsampling_steps == 2
    cfg2 = Config.from_yaml(str(y), mode="test", overrides={"generation": {"upsampling_steps": 0}})
    assert cfg2.config["mode"] == "test" and cfg2.config["generation"]["upsampling_steps"] == 0


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference checkout not present")
def test_the_reference_config_file_itself_loads():
    from rfdnet_amd.iscnet.config import Config, DEFAULT_CONFIG
    cfg = Config.from_yaml(REF_YAML, mode="demo")
    for sect in ("data", "generation"):
        for k, v in DEFAULT_CONFIG[sect].items():
            assert cfg.config[sect][k] == v, (sect, k)       # our defaults ARE that file's values
    assert cfg.eval_overrides["cls_nms"] is True and cfg.eval_overrides["remove_empty_box"] is True


def test_worker_view_shares_every_weight_but_not_the_generator_state():
    """ISCNet.worker_view(): the handle bench.py gives each in-flight scene -- one set of parameters / buffers per GPU
    (identical tensor objects, identical state_dict keys), a generator of its own (statistics, mesh buffers, round
    hook), and re-pointing the view's `completion` leaves the base network untouched."""
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    net = ISCNet(Config({'generation': {'resolution_0': 8, 'upsampling_steps': 1}}))
    view = net.worker_view()
    base_params = dict(net.named_parameters())
    view_params = dict(view.named_parameters())
    assert list(base_params) == list(view_params)
    assert all(view_params[k] is base_params[k] for k in base_params)
    assert all(a is b for a, b in zip(net.buffers(), view.buffers()))
    assert list(net.state_dict()) == list(view.state_dict())
    assert view.completion is not net.completion and view.completion.decoder is net.completion.decoder
    g0, g1 = net.completion.generator, view.completion.generator
    assert g1 is not g0 and g1.model is view.completion and g0.model is net.completion
    assert (g1.resolution0, g1.upsampling_steps, g1.threshold) == (g0.resolution0, g0.upsampling_steps, g0.threshold)
    g1.stats['n_queries'] = 5
    g1.round_hook = lambda r, d: None
    assert g0.stats == {} and g0.round_hook is None
    assert net._modules['completion'] is net.completion and net.completion.generator is g0


def test_a_scene_flagged_at_the_old_scale_reruns_when_another_scene_lowered_the_shared_decoder(monkeypatch):
    """every scene in flight shares ONE decoder (worker_view), so its activation scale may be lowered by
    scene A while scene B -- folded at the old scale -- is still running.  B's range flag must then be answered by a
    re-run at the fallback scale, although lower_activation_scale() says "already there"; only a flag raised by a run
    that itself used the fallback scale is a real overflow and raises.  (CPU: the status word and the runs are stubs.)"""
    import types
    import pytest
    from rfdnet_amd import _lib
    from rfdnet_amd.iscnet.network import ISCNet

    class Dec(object):
        ka = 6

        def lower_activation_scale(self):
            if self.ka <= 3:
                return False
            self.ka = 3
            return True

    def make(statuses, other_scene_lowers):
        dec, runs = Dec(), []

        def run(codes, cls):
            runs.append(dec.ka)
            if other_scene_lowers and len(runs) == 1:
                dec.ka = 3                      # scene A, on another host thread, answers ITS flag meanwhile
            return "meshes@%d" % runs[-1]
        gen = types.SimpleNamespace(generate_mesh=run, generate_grids=run)
        net = types.SimpleNamespace(completion=types.SimpleNamespace(generator=gen, decoder=dec))
        seq = list(statuses)
        monkeypatch.setattr(_lib, "stream_status_bits", lambda: seq.pop(0))
        import contextlib
        import torch
        monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())    # no GPU in this tier
        return net, runs

    net, runs = make([2, 0], other_scene_lowers=True)
    assert ISCNet.complete(net, None, None, None) == "meshes@3" and runs == [6, 3]
    net, runs = make([2, 0], other_scene_lowers=False)          # the ordinary case: this scene lowers the scale itself
    assert ISCNet.complete(net, None, None, None) == "meshes@3" and runs == [6, 3]
    net, runs = make([2], other_scene_lowers=False)
    net.completion.decoder.ka = 3                               # flagged AT the fallback scale: a real overflow
    with pytest.raises(_lib.RfdHipError, match="occupancy decoder"):
        ISCNet.complete(net, None, None, None)
    assert runs == [3]
    net, runs = make([2, 2], other_scene_lowers=True)           # still flagged after the re-run: raises
    with pytest.raises(_lib.RfdHipError, match="occupancy decoder"):
        ISCNet.complete(net, None, None, None)
    assert runs == [6, 3]
