#!/usr/bin/env python3
"""Render a task list with the command-line renderer and write an HTML + JSON report.

Counterpart of the reference's ci/gen_report.py:149-247 without the uploads: for every task of the
YAML config and every render config of the task, run
    <executable> --scene-config S --render-config R --output <out>/<task>/<R.stem>.png
wall-clock the whole invocation (ci/gen_report.py:188-201), keep its log, and list image, caption
"<task>@<sampler> Sampler x <spp>spp@<W> x <H>" and time cost in report.html.  report.json adds the
renderer's own counters (samples, rays, kernel time) when the executable writes them (--stats-json).

    python ci/gen_report.py --bin rustraytracer_amd/host/rrt --config ci/report_configs.yml --output_dir report

Takes the reference harness's command line unchanged (ci/gen_report.py:250-258): --bin, --config, --output_dir, --id,
--upload, --no_compress.  Without --no_compress every PNG is also converted to a JPEG, which the report then shows
(ci/gen_report.py:203-211); --id is recorded in report.json; --upload is accepted and nothing is uploaded (the
reference posts to imgur and a storage bucket; this harness has no network side).  --executable and --output-dir are
this harness's own spellings of --bin and --output_dir.
"""
import argparse
import html
import json
import subprocess
import sys
import time
from pathlib import Path

import yaml

try:
    import tomllib as toml
except ModuleNotFoundError:
    import tomli as toml


def git_head():
    try:
        h = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        d = subprocess.run(["git", "log", "-1", "--pretty=%B"], capture_output=True, text=True, check=True).stdout.strip()
        return h, d
    except Exception:
        return "unknown", ""


def convert_png_to_jpg(png_path, jpg_path):
    """ci/gen_report.py:60-66 of the reference: RGB JPEG next to the PNG."""
    from PIL import Image

    with Image.open(png_path) as im:
        im.convert("RGB").save(jpg_path, "JPEG", quality=95)


_STATS_FLAG = {}


def takes_stats_json(executable):
    """Whether the binary knows --stats-json: asked once through its --help.  The reference's own binary has exactly the
    three flags -s / -r / -o (src/main.rs:25-34) and would stop at an unknown one, so it is run with those alone."""
    key = str(executable)
    if key not in _STATS_FLAG:
        try:
            r = subprocess.run([key, "--help"], capture_output=True, text=True, timeout=60)
            _STATS_FLAG[key] = "--stats-json" in (r.stdout + r.stderr)
        except Exception:  # noqa: BLE001
            _STATS_FLAG[key] = False
    return _STATS_FLAG[key]


def run_task(executable, scene_config, render_config, out_dir, extra_args, compress=False):
    stem = Path(render_config).stem
    image, log, stats = out_dir / f"{stem}.png", out_dir / f"{stem}.log", out_dir / f"{stem}.json"
    cmd = [str(executable), "--scene-config", str(scene_config), "--render-config", str(render_config), "--output",
           str(image), *(["--stats-json", str(stats)] if takes_stats_json(executable) else []), *extra_args]
    t0 = time.perf_counter()
    with open(log, "w") as lf:
        proc = subprocess.run(cmd, stdout=lf, stderr=subprocess.STDOUT)
    elapsed = time.perf_counter() - t0
    rc = toml.load(open(render_config, "rb"))
    entry = {
        "render_config": str(render_config),
        "image": image.name if proc.returncode == 0 else None,
        "log": log.name,
        "returncode": proc.returncode,
        "time_cost_s": elapsed,
        "width": rc["image"]["width"],
        "height": rc["image"]["height"],
        "sampler": rc["sampler"]["type"],
        "spp": rc["sampler"]["samples_per_pixel"],
    }
    if proc.returncode == 0 and compress:
        jpg = out_dir / f"{stem}.jpg"
        convert_png_to_jpg(image, jpg)
        entry["image"] = jpg.name
        entry["png"] = image.name
    if proc.returncode == 0 and stats.exists():
        st = json.load(open(stats))
        entry["stats"] = st
        entry["msamples_per_s_wall"] = st["samples"] / elapsed / 1e6
        entry["msamples_per_s_render"] = st["samples"] / st["render_s"] / 1e6
        entry["mrays_per_s_render"] = st["rays"] / st["render_s"] / 1e6
    return entry


def report_html(rep):
    parts = []
    for task in rep["tasks"]:
        rows = []
        for e in task["renders"]:
            cap = f"{task['name']}@{e['sampler']} Sampler x {e['spp']}spp@{e['width']} x {e['height']}"
            img = (f'<img src="{html.escape(task["dir"])}/{html.escape(e["image"])}" style="max-width: 480px">'
                   if e["image"] else f"<b>failed (exit {e['returncode']})</b>")
            rate = f", {e['msamples_per_s_render']:.1f} Msamples/s" if "msamples_per_s_render" in e else ""
            rows.append(f"<figure>{img}<figcaption>{html.escape(cap)}<br>time cost: {e['time_cost_s']:.2f} s{rate} "
                        f'(<a href="{html.escape(task["dir"])}/{html.escape(e["log"])}">log</a>)</figcaption></figure>')
        parts.append(f"<h2>{html.escape(task['name'])}</h2>\n" + "\n".join(rows))
    return (f"<html><head><meta charset=\"utf-8\"><title>Render report</title></head><body><h1>Render report</h1>"
            f"<p>Commit: {html.escape(rep['commit_hash'])}</p><p>Commit description: {html.escape(rep['commit_description'])}</p>"
            + "\n".join(parts) + "</body></html>\n")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--bin", "--executable", dest="executable", type=Path, required=True)
    ap.add_argument("--config", type=Path, required=True)
    ap.add_argument("--output_dir", "--output-dir", dest="output_dir", type=Path, required=True)
    ap.add_argument("--id", type=str, default=None, help="report id, recorded in report.json")
    ap.add_argument("--upload", action="store_true", help="accepted for compatibility; nothing is uploaded")
    ap.add_argument("--no_compress", "--no-compress", dest="no_compress", action="store_true",
                    help="keep the PNGs only (default: also write a JPEG per render and show that)")
    ap.add_argument("--extra-arg", action="append", default=[], help="passed through to the executable (repeatable)")
    args = ap.parse_args(argv)
    cfg = yaml.safe_load(open(args.config, "r", encoding="utf-8"))
    commit, desc = git_head()
    rep = {"id": args.id, "commit_hash": commit, "commit_description": desc, "tasks": []}
    if args.upload:
        print("--upload: this harness uploads nothing; images and logs stay under the output directory", flush=True)
    args.output_dir.mkdir(parents=True, exist_ok=True)
    failed = 0
    for task in cfg["tasks"]:
        tdir = args.output_dir / task["name"]
        tdir.mkdir(parents=True, exist_ok=True)
        t = {"name": task["name"], "dir": task["name"], "scene_config": task["scene_config"], "renders": []}
        for rc in task["render_configs"]:
            e = run_task(args.executable, Path(task["scene_config"]), Path(rc), tdir, args.extra_arg, compress=not args.no_compress)
            failed += e["returncode"] != 0
            t["renders"].append(e)
            print(f"{task['name']}: {rc}: exit {e['returncode']}, {e['time_cost_s']:.2f} s", flush=True)
        rep["tasks"].append(t)
    (args.output_dir / "report.html").write_text(report_html(rep), encoding="utf-8")
    (args.output_dir / "report.json").write_text(json.dumps(rep, indent=1), encoding="utf-8")
    print(f"report generated at {args.output_dir}", flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
