"""Time pmx_emit_team_obs (uint8 planes) on the bench workloads: python tools/emit_ab.py [workload ...]  (GPU box).
(profiles/r03/emit_team_ab.txt was written with a scratch build that still held round 2's kernel behind PMX_EMIT_PER_SLOT=1.)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for name in (sys.argv[1:] or ["small16384", "blox4096", "mazes8192"]):
        import pmx
        layname, n_envs = bench.WORKLOADS[name]
        if layname == "mazeGenerator":
            from pmx import maze_generator
            lay = [pmx.Layout.from_text(maze_generator.generate_maze(seed)) for seed in range(1, n_envs + 1)]
        else:
            lay = pmx.get_layout(layname)
        length = 300
        g = torch.Generator(device=dev).manual_seed(1234)
        actions = torch.randint(0, 5, (64, n_envs, 4), generator=g, device=dev, dtype=torch.int8)
        r = bench.emit_team_probe(lay, n_envs, length, dev, 0, actions)
        print(json.dumps({"workload": name, "us": round(r["avg_launch_us"], 2),
                          "frac": round(r["frac"], 3), "checksum": r["checksum"]}), flush=True)
