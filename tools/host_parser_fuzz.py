"""Mutation fuzz of the C++ host's parsers (CPU only): host_parser_fuzz.py [iterations, default 3000] [seed]
Valid inputs -- the shipped scene and render TOML files, a binary and an ASCII PLY mesh -- with random bytes changed,
removed, doubled or cut off, through `rrt --dump-scene` (scene + meshes; no GPU call) and through the render-config
reader, on a build with AddressSanitizer + UBSan (tests/c_abi/rrt_san, built by tests/test_host_cli.py).  Every run must
end in exit status 0, 2 or 101 (where the reference panics): never a signal, a sanitizer report or a hang."""
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "c_abi", "rrt_san")


def mutate(data, rng):
    data = bytearray(data)
    for _ in range(rng.choice([1, 1, 2, 4, 16])):
        if not data:
            break
        i = rng.randrange(len(data))
        op = rng.randrange(6)
        if op == 0:
            data[i] = rng.randrange(256)
        elif op == 1:
            del data[i:i + rng.choice([1, 1, 4, 64])]
        elif op == 2:
            data[i:i] = data[i:i + rng.choice([1, 8, 64])]
        elif op == 3:
            del data[i:]
        elif op == 4:
            data[i:i] = rng.choice([b"-", b"9999999999999999999999", b"1e999", b"nan", b"\n[[objects]]\n", b"{", b"}", b"\"", b"=", b"\x00", b"4294967295 "])
        else:
            data[i] = data[i] ^ (1 << rng.randrange(8))
    return bytes(data)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    if not os.path.exists(SAN):
        sys.exit("build tests/c_abi/rrt_san first (python -m pytest tests/test_host_cli.py -k sanitizers)")
    work = tempfile.mkdtemp(prefix="rrt_fuzz_")
    shutil.copytree(os.path.join(ROOT, "configs"), os.path.join(work, "configs"))
    scene = os.path.join(work, "configs", "cornell_box", "scene.toml")
    render = os.path.join(work, "configs", "cornell_box", "render_low_quality.toml")
    ply = os.path.join(work, "configs", "cornell_box", "cornell_box_0000_m000.ply")
    originals = {p: open(p, "rb").read() for p in (scene, render, ply)}
    # an ASCII variant of the mesh (the loader reads both)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:abort_on_error=1")
    outcomes, bad = {}, 0
    for it in range(n):
        target = rng.choice([scene, scene, render, ply, ply])
        for p, d in originals.items():
            open(p, "wb").write(d)
        open(target, "wb").write(mutate(originals[target], rng))
        if target == render:  # the render config is read before any GPU call; without a GPU the run then ends with 101
            cmd = [SAN, "-s", scene, "-r", render, "-o", os.path.join(work, "o.png")]
        else:
            cmd = [SAN, "-s", scene, "--dump-scene", os.path.join(work, "dump.bin")]
        try:
            r = subprocess.run(cmd, cwd=work, env=env, capture_output=True, timeout=60)
            code = r.returncode
        except subprocess.TimeoutExpired:
            code = "timeout"
        outcomes[code] = outcomes.get(code, 0) + 1
        if code not in (0, 2, 101):
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_case_%d_%s" % (it, os.path.basename(target)))
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            shutil.copy(target, keep)
            print("iteration %d: %s on %s -> %s\n%s" % (it, os.path.basename(target), " ".join(cmd[1:3]), code, r.stderr.decode(errors="replace")[-1500:] if code != "timeout" else ""), flush=True)
    shutil.rmtree(work, ignore_errors=True)
    print("outcomes:", outcomes, "problems:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
