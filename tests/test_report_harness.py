"""ci/gen_report.py takes the command line of the reference's harness unchanged (ci/gen_report.py:250-258 there):
--bin, --config, --output_dir, --id, --upload, --no_compress.  A stand-in renderer (a script that honours the
renderer's three flags and writes a PNG) keeps this test off the GPU; tests/test_gpu_cli.py drives the real binary."""
import json
import os
import stat
import subprocess
import sys

from conftest import ROOT

FAKE = '''#!%s
import argparse, json
from PIL import Image
ap = argparse.ArgumentParser()
ap.add_argument("--scene-config"); ap.add_argument("--render-config"); ap.add_argument("--output"); ap.add_argument("--stats-json")
a = ap.parse_args()
Image.new("RGB", (8, 6), (200, 100, 50)).save(a.output)
json.dump({"samples": 48, "rays": 100, "render_s": 0.5}, open(a.stats_json, "w"))
print("rendered", a.output)
'''

TASKS = '''tasks:
- name: "cornell box"
  scene_config: "configs/cornell_box/scene.toml"
  render_configs:
  - "configs/cornell_box/render_low_quality.toml"
'''


def _run(tmp_path, *flags):
    fake = tmp_path / "fake_rrt"
    fake.write_text(FAKE % sys.executable)
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    cfg = tmp_path / "tasks.yml"
    cfg.write_text(TASKS)
    out = tmp_path / "report"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "ci", "gen_report.py"), "--bin", str(fake), "--config", str(cfg),
                        "--output_dir", str(out), *flags], capture_output=True, text=True, timeout=120, cwd=ROOT)
    return r, out


def test_reference_flag_names_no_compress(tmp_path):
    r, out = _run(tmp_path, "--id", "abc123", "--no_compress")
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.load(open(out / "report.json"))
    assert rep["id"] == "abc123"
    e = rep["tasks"][0]["renders"][0]
    assert e["returncode"] == 0 and e["image"] == "render_low_quality.png" and (out / "cornell box" / e["image"]).exists()
    assert not (out / "cornell box" / "render_low_quality.jpg").exists()
    html = (out / "report.html").read_text()
    assert "cornell box@" in html and "Sampler x" in html and "time cost" in html


def test_reference_flag_names_compress_and_upload(tmp_path):
    r, out = _run(tmp_path, "--upload")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "uploads nothing" in r.stdout
    e = json.load(open(out / "report.json"))["tasks"][0]["renders"][0]
    assert e["image"] == "render_low_quality.jpg" and e["png"] == "render_low_quality.png"
    assert (out / "cornell box" / e["image"]).exists() and (out / "cornell box" / e["png"]).exists()
    assert "render_low_quality.jpg" in (out / "report.html").read_text()


THREE_FLAGS = '''#!%s
import sys
from PIL import Image
# the reference's command line: exactly -s / -r / -o (src/main.rs:25-34), no --help text that names other flags,
# and anything else is an error
args = sys.argv[1:]
if args == ["--help"]:
    print("Usage: rust-ray-tracer --scene-config <SCENE_CONFIG> --render-config <RENDER_CONFIG> --output <OUTPUT>")
    sys.exit(0)
known = {"--scene-config", "--render-config", "--output", "-s", "-r", "-o"}
opts = dict(zip(args[0::2], args[1::2]))
if len(args) != 6 or set(opts) - known:
    print("error: unexpected argument", file=sys.stderr)
    sys.exit(2)
Image.new("RGB", (8, 6), (10, 20, 30)).save(opts.get("--output") or opts["-o"])
'''


def test_a_binary_with_only_the_reference_flags_runs_unchanged(tmp_path):
    """--bin pointing at a renderer that has the reference's three flags and nothing else: the harness must not hand it
    a flag of this build's own."""
    fake = tmp_path / "three_flag_rrt"
    fake.write_text(THREE_FLAGS % sys.executable)
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    cfg = tmp_path / "tasks.yml"
    cfg.write_text(TASKS)
    out = tmp_path / "report"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "ci", "gen_report.py"), "--bin", str(fake), "--config", str(cfg),
                        "--output_dir", str(out), "--no_compress"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    e = json.load(open(out / "report.json"))["tasks"][0]["renders"][0]
    assert e["returncode"] == 0 and e["image"] == "render_low_quality.png" and "stats" not in e
