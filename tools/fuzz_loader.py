"""Mutates generated scene directories (scenes.random_box) -- JSON keys dropped or retyped, files truncated, bytes of OBJ
and map files overwritten -- and loads each with the host loader: it must load or fail with an error message, never
crash.  usage: python tools/fuzz_loader.py <first_seed> <last_seed>; exit status 0 when no load crashed.
With PTK_FUZZ_LOADER_EXE set to a build of tests/cpp/load_scene_main.cpp (tools/sanitize_host.sh builds it with the
address and undefined-behaviour sanitizers), every directory is loaded by that program in a process of its own: any
exit status but 0, or output that is neither "loaded" nor "error:", counts as a crash."""
import subprocess
import json
import os
import random
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as pt
from pathtracing_amd import scenes

def mutate(node, rnd, depth=0):
    """one random structural mutation somewhere in a JSON tree"""
    if isinstance(node, dict) and node:
        k = rnd.choice(list(node.keys()))
        r = rnd.random()
        if r < 0.25: del node[k]
        elif r < 0.5: node[k] = rnd.choice([None, "x", -1, 1e308, [], {}, [1, 2], "nan", True, 3.5, "", [0, 0, 0, 0]])
        else: mutate(node[k], rnd, depth + 1) if isinstance(node[k], (dict, list)) else node.__setitem__(k, rnd.choice([None, "x", -1, [], {}, 0, 1e308, "missing_name"]))
    elif isinstance(node, list) and node:
        i = rnd.randrange(len(node))
        r = rnd.random()
        if r < 0.2: del node[i]
        elif r < 0.4: node[i] = rnd.choice([None, "x", -1, {}, []])
        elif isinstance(node[i], (dict, list)): mutate(node[i], rnd, depth + 1)
        else: node[i] = rnd.choice([None, "x", {}, [], 1e308])

first, last = int(sys.argv[1]), int(sys.argv[2])
ok = err = 0
for seed in range(first, last):
    rnd = random.Random(seed)
    d = tempfile.mkdtemp(prefix="ptk_fuzz_")
    scenes.random_box(d, seed % 40, extras=True, WIDTH=8, HEIGHT=8)
    which = rnd.choice(["scene.json", "scene.json", "settings.json", "obj", "map", "truncate"])
    if which in ("scene.json", "settings.json"):
        p = os.path.join(d, which)
        tree = json.load(open(p))
        for _ in range(rnd.randint(1, 3)): mutate(tree, rnd)
        json.dump(tree, open(p, "w"))
    elif which == "truncate":
        p = os.path.join(d, "scene.json")
        data = open(p).read()
        open(p, "w").write(data[:rnd.randrange(len(data))])
    else:
        sub = "models" if which == "obj" else "maps"
        files = [os.path.join(d, sub, f) for f in os.listdir(os.path.join(d, sub))] if os.path.isdir(os.path.join(d, sub)) else []
        if files:
            p = rnd.choice(files)
            data = bytearray(open(p, "rb").read())
            for _ in range(rnd.randint(1, 20)):
                if data: data[rnd.randrange(len(data))] = rnd.randrange(256)
            if rnd.random() < 0.3: data = data[:rnd.randrange(len(data) + 1)]
            open(p, "wb").write(bytes(data))
    print(seed, which, end=" ", flush=True)
    exe = os.environ.get("PTK_FUZZ_LOADER_EXE")
    if exe:
        r = subprocess.run([exe, d], capture_output=True, text=True, timeout=120)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
        if r.returncode != 0 or not (line.startswith("loaded") or line.startswith("error:")):
            print("CRASH, exit status", r.returncode, flush=True)
            print(r.stdout[-2000:], r.stderr[-6000:], flush=True)
            sys.exit(1)
        if line.startswith("loaded"): ok += 1
        else: err += 1
        print(line[:80], flush=True)
    else:
        try:
            pt.load_scene_directory(d)
            ok += 1; print("loaded", flush=True)
        except pt.PtkError as e:
            err += 1; print("error:", str(e)[:80], flush=True)
    shutil.rmtree(d, ignore_errors=True)
print("loaded", ok, "refused", err)
