"""bench.py's host-side contract, without a GPU: the self-launch of N ranks, the kernel-source fingerprint the traffic figure
is keyed on, and the workload table."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` outside torch.distributed.run starts N ranks as child processes under torch.distributed.run on
    127.0.0.1 with its own arguments passed through, and returns their exit code -- before anything touches the GPU."""
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--config", "lego_b64"])
    args = bench.parse_args()
    assert args.gpus == 4 and args.config == "lego_b64"
    assert bench.relaunch_as_ranks(args) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "5", "--config", "lego_b64"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_defaults_and_workloads():
    import bench
    from iffnerf_amd import synthetic
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse_args()
    finally:
        sys.argv = old
    assert (a.gpus, a.config, a.in_flight) == (1, "lego16k", 4) and a.steps >= 100 and a.warmup >= 10
    assert set(synthetic.WORKLOADS) == {"lego16k", "truck32k", "bicycle64k", "lego_b64", "lego540k"}
    assert synthetic.WORKLOADS["lego540k"]["gen_points"] == 20000          # the reference's default (model_utils.py:22)
    assert synthetic.WORKLOADS["lego16k"]["gen_points"] * 27 == 16011
    assert synthetic.WORKLOADS["truck32k"]["gen_points"] * 27 == 32022
    assert synthetic.WORKLOADS["bicycle64k"]["gen_points"] * 27 == 64017
    assert synthetic.WORKLOADS["lego_b64"]["queries"] == 64 and synthetic.WORKLOADS["lego_b64"]["shared_rays"]


def test_traffic_figure_is_keyed_on_the_kernel_sources():
    """profiles/r06_hbm_traffic_<config>.json carries the fingerprint of the sources it was measured on and the workload; bench.py
    and the summariser compute the fingerprint the same way (bench.py reports `roofline.traffic` / `frac_hbm_counters` only while both
    agree -- it takes the newest rNN file of its --config)."""
    import bench
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import summarize_pmc
    fp = bench.source_fingerprint()
    assert fp == summarize_pmc.source_fingerprint() and len(fp) == 16
    for cfg, march in (("lego16k", "k4f_fan_march<3>"), ("truck32k", "k4f_fan_march<3>"), ("bicycle64k", "k4g_fan_march<22, 3, 3>"),
                       ("lego_b64", "k4f_fan_march<3>"), ("lego540k", "k4f_fan_march<3>")):
        j = json.load(open(os.path.join(ROOT, "profiles", f"r06_hbm_traffic_{cfg}.json")))
        assert j["config"] == cfg and len(j["source_sha16"]) == 16
        for k in ("k5_trunk_h<1, 1, 2>", march):
            assert j["kernels"][k]["hbm_bytes_per_launch"] > 0 and j["kernels"][k]["valu_wave_insts"] > 0, (cfg, k)
        # the committed counters belong to the kernel sources in the tree: the driver's bench line can carry them
        assert j["source_sha16"] == fp, f"profiles/r06_hbm_traffic_{cfg}.json was collected on other kernel sources: re-run scripts/gpu_profile.sh"


def test_roofline_object_states_both_hbm_readings():
    """The march's roofline object prints SURVEY 8(d)'s own fraction (algorithmic bytes / time / 8 TB/s: above 1 for the fused fan
    kernel, labelled as not a bound) and the HBM fraction from the counters beside the LDS-model `frac` -- round 4's numbers in."""
    import bench
    e = bench.gather_kernel_entry("k4f_fan_march<3>", 512352, "lds-gather", bench.LDS_PEAK_GBS, 38260922573, 1.0406, 1284353698, {"lds_busy": 0.6},
                                  "note", "basis")
    assert abs(e["frac_8d"] - 4.596) < 0.01 and "not" in e["frac_8d_note"].lower()
    assert abs(e["frac_hbm_counters"] - 0.1543) < 0.001 and abs(e["frac"] - 0.2338) < 0.001
    assert e["traffic"] == 1284353698 and e["bound"] == "lds-gather" and e["duration_source"].startswith("hipEvents")
    stale = bench.gather_kernel_entry("k", 1, "l1-gather", bench.L1_PEAK_GBS, 1e9, 1.0, None, None, "", "")
    assert stale["frac_hbm_counters"] is None and stale["traffic"] is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("dropin_poses_per_s", "cold_image_to_pose_per_s", "value_without_settle", "dev_library"):
        assert '"%s"' % key in src, key
    # the route users run carries its own roof: the logits round trip of the kept rows at the HBM rate over the loop's time per image
    r = bench.dropin_roofline(540000, 256.0, 256.0, 0.49e-3, 3)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["logits_bytes_per_image"] == 2 * 256 * 540000 * 4
    assert abs(r["frac"] - 1105920000 / 0.49e-3 / 8e12) < 1e-3 and abs(r["floor_ms_per_image"] - 0.1382) < 1e-3
    assert abs(r["mfma"]["frac_issued"] - 256 * 540000 * 3 * 512 / 0.49e-3 / 2.5e15) < 1e-3 and r["mfma"]["kernel"] == "iff_logits_from_cache_rows"
    assert '"roofline": dropin_roofline(' in src
    # the one-product class is a labelled extra with its accuracy beside it, never the default arithmetic and never `value`
    assert 'result["fast_class"] = fast_class_report(' in src and '"--gemm", "f16x1"' in src
    from iffnerf_amd import hip_identify as H
    assert H.GEMM_DEFAULT == H.GEMM_F16X2 and H.GEMM_F16X1 == 4
