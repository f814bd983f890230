"""Calibration artefacts: the reference's ./saved file protocol, the three tensor rules
that produce it, validation, and the quantised-weight cache."""
import math

import pytest
import torch

from arcquant_amd import artifacts as A
from tests.util import outlier_activations


def _stats(k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(k, generator=g) * 10


def test_file_names_follow_the_reference():
    # reorder_indices.py:88-90, model/main.py:112-114
    assert A.artifact_path("reorder_index", "Llama-3.1-8B", "WikiText2", "max") == \
        "./saved/llama-3.1-8b_reorder_index_wikitext2_max.pt"
    assert A.artifact_path("select_num", "Qwen2.5-7B", "pile", "hessian", root="/x") == "/x/qwen2.5-7b_select_num_pile_hessian.pt"
    assert A.layer_input_name(3, "mlp", "down_proj") == "layers.3.mlp.down_proj.input"
    with pytest.raises(ValueError):
        A.artifact_path("weights", "m", "d", "max")


def test_reorder_index_puts_outliers_last():
    stat = torch.tensor([0.5, 9.0, 0.1, 3.0])
    idx = A.reorder_index_from_stat(stat)
    assert idx.tolist() == [2, 0, 3, 1]
    with pytest.raises(ValueError):
        A.reorder_index_from_stat(torch.zeros(2, 2))


def test_channel_stat_is_a_running_max_of_magnitudes():
    a = torch.tensor([[1.0, -5.0], [-2.0, 3.0]])
    b = torch.tensor([[[-4.0, 1.0]]])
    s = A.channel_stat(a)
    assert s.tolist() == [2.0, 5.0]
    assert A.channel_stat(b, s).tolist() == [4.0, 5.0]


def test_select_num_rule():
    # utilize.py:465-478 restated by hand on a case with a known count
    k, rows = 256, 8
    x = torch.full((rows, k), 0.01)
    x[:, :10] = 1.0                      # 10 channels above max/8 in every row
    idx = A.reorder_index_from_stat(A.channel_stat(x))
    ke, bits = A.select_num_from_samples(x, idx)
    assert ke == math.ceil(k * (10 / k) / 64) * 64 == 64
    assert bits == pytest.approx(4.5 * (k + 64) / k)
    # the comparison is on signed values: a large negative entry is not selected
    y = x.clone()
    y[:, :10] = -1.0
    assert A.select_num_from_samples(y, idx)[0] == 256          # max is 0.01, every 0.01 exceeds 0.00125
    ke2, _ = A.select_num_from_samples(outlier_activations(64, 4096, seed=2).float(), torch.arange(4096))
    assert ke2 % 64 == 0 and 0 < ke2 <= 4096


def test_save_load_round_trip(tmp_path):
    names = list(A.names_for_decoder(2))
    assert len(names) == 14 and names[0] == "layers.0.self_attn.q_proj.input"
    stats = {n: _stats(256, i) for i, n in enumerate(names)}
    stats["layers.0.mlp.down_proj.output"] = _stats(256, 99)       # ignored like the reference does
    samples = {names[0]: outlier_activations(16, 256, seed=1).float()}
    cal = A.calibration_from_stats(stats, samples, default_select_num=64)
    assert set(cal.reorder_index) == set(names)
    assert cal.select_num[names[1]] == 64
    paths = A.save_calibration(cal, "Toy-1B", "Synthetic", "max", root=str(tmp_path))
    assert set(paths) == set(A.KINDS)
    back = A.load_calibration("toy-1b", "synthetic", "max", root=str(tmp_path), require_act_scales=True)
    for n in names:
        assert torch.equal(back.reorder_index[n], cal.reorder_index[n])
        assert back.select_num[n] == cal.select_num[n]
        assert torch.equal(back.act_scales[n], cal.act_scales[n])
    assert back.total_average_bits() == pytest.approx(cal.total_average_bits())
    assert back.device_index(names[0], "cpu").dtype == torch.int16
    # the files are plain dicts a reference checkout can torch.load
    raw = torch.load(paths["select_num"], weights_only=True)
    assert raw[names[1]] == 64


def test_missing_and_corrupt_files_are_refused(tmp_path):
    with pytest.raises(FileNotFoundError):
        A.load_calibration("none", "d", "max", root=str(tmp_path))
    good = {"a.input": torch.randperm(128)}
    torch.save(good, A.artifact_path("reorder_index", "m", "d", "max", str(tmp_path)))
    torch.save({"a.input": 96}, A.artifact_path("select_num", "m", "d", "max", str(tmp_path)))
    with pytest.raises(ValueError, match="multiple of 64"):
        A.load_calibration("m", "d", "max", root=str(tmp_path))
    torch.save({"a.input": 192}, A.artifact_path("select_num", "m", "d", "max", str(tmp_path)))
    with pytest.raises(ValueError, match="multiple of 64"):                    # KE > K
        A.load_calibration("m", "d", "max", root=str(tmp_path))
    torch.save({"a.input": 64}, A.artifact_path("select_num", "m", "d", "max", str(tmp_path)))
    assert A.load_calibration("m", "d", "max", root=str(tmp_path)).select_num["a.input"] == 64
    bad = torch.randperm(128)
    bad[5] = bad[6]
    torch.save({"a.input": bad}, A.artifact_path("reorder_index", "m", "d", "max", str(tmp_path)))
    with pytest.raises(ValueError, match="not a permutation"):
        A.load_calibration("m", "d", "max", root=str(tmp_path))
    with pytest.raises(ValueError):
        A.check_reorder_index("x", torch.rand(8))
    with pytest.raises(ValueError, match="int16"):
        A.check_reorder_index("x", torch.arange(40000))
    torch.save([1, 2], A.artifact_path("reorder_index", "m", "d", "max", str(tmp_path)))
    with pytest.raises(ValueError, match="dict"):
        A.load_calibration("m", "d", "max", root=str(tmp_path))


@pytest.mark.gpu
def test_quantized_weight_cache_round_trip(tmp_path):
    from arcquant_amd import qlinear
    torch.manual_seed(0)
    dev = "cuda"
    idx = torch.randperm(256)
    layers = {}
    for name, bias in (("layers.0.mlp.up_proj", False), ("layers.0.self_attn.q_proj", True)):
        lin = torch.nn.Linear(256, 384, bias=bias, dtype=torch.bfloat16, device=dev)
        layers[name] = qlinear.QLinearLayer(lin, 64, idx)
    path = str(tmp_path / "w.pt")
    A.save_quantized_weights(layers, path)
    entries = A.load_quantized_weights(path, device=dev)
    x = outlier_activations(4, 256, seed=3).to(dev)
    qx = qlinear.NVFP4_reorder_quantize_x(x, idx.to(dev, torch.int16), 64)
    for name, m in layers.items():
        r = A.restore_qlinear(entries[name])
        assert torch.equal(r.W, m.W) and torch.equal(r.scale_w, m.scale_w)
        assert torch.equal(r((*qx, 1, 4)), m((*qx, 1, 4)))
    # a truncated scale buffer is refused before it can reach the GEMM
    blob = torch.load(path, weights_only=True)
    blob["layers.0.mlp.up_proj"]["scale_w"] = blob["layers.0.mlp.up_proj"]["scale_w"][:-1]
    torch.save(blob, path)
    with pytest.raises(ValueError, match="scale buffer"):
        A.load_quantized_weights(path, device=dev)
