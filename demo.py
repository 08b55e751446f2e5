#!/usr/bin/env python
"""`python main.py --mode demo` of the reference (main.py:17-38, demo.py:379-420)
for the MI355X path: load a scan (.off) or a seeded synthetic scene, reconstruct,
write the reference's output files.

  python demo.py --demo_path demo/inputs/scene0549_00.off --out out/scene0549_00
  python demo.py --synthetic 10 --out out/synth --upsampling_steps 1

Weights: --weight <pretrained_weight.pth> (reference checkpoint, key names kept);
without it seeded random weights are used (no pretrained weights ship with the
reference checkout)."""
import argparse
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default=None,
                    help="a reference config file (configs/config_files/ISCNet_test.yaml); its data / model / test / "
                         "generation blocks and its `weight` list are used, flags below override it")
    ap.add_argument("--mode", choices=["demo", "test"], default="demo")
    ap.add_argument("--demo_path", type=str, default=None)
    ap.add_argument("--synthetic", type=int, default=None, help="seed of a synthetic ScanNet-like scene")
    ap.add_argument("--weight", type=str, default=None)
    ap.add_argument("--out", type=str, default="out/demo")
    ap.add_argument("--resolution_0", type=int, default=None)
    ap.add_argument("--upsampling_steps", type=int, default=None)       # ISCNet_test.yaml:62-63 (32, 0)
    ap.add_argument("--selection", choices=["nms", "all", "objectness"], default="nms")
    ap.add_argument("--mean_size_npz", type=str, default=None,
                    help="class mean sizes (the reference's datasets/scannet/scannet_means.npz); default: "
                         "$RFD_MEAN_SIZE_NPZ or that path relative to the working directory")
    ap.add_argument("--allow-placeholder-sizes", action="store_true",
                    help="synthetic runs only: let --selection nms decode boxes with placeholder class mean sizes when "
                         "no scannet_means.npz is available (the reference fails hard on the missing file; so does this "
                         "path without the flag)")
    args = ap.parse_args()

    from rfdnet_amd import io, synthetic
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet

    gen = {k: v for k, v in (('resolution_0', args.resolution_0), ('upsampling_steps', args.upsampling_steps))
           if v is not None}
    if args.config:
        cfg = Config.from_yaml(args.config, mode=args.mode, overrides={'generation': gen},
                               mean_size_arr=args.mean_size_npz)
        if not args.weight and cfg.config.get('weight'):
            import os
            w = cfg.config['weight'][0] if isinstance(cfg.config['weight'], (list, tuple)) else cfg.config['weight']
            if os.path.exists(w):
                args.weight = w
    else:
        cfg = Config({'generation': gen}, mean_size_arr=args.mean_size_npz)
    if args.allow_placeholder_sizes:
        cfg.eval_overrides['allow_placeholder_sizes'] = True
    net = ISCNet(cfg)
    if args.weight:
        ckpt = torch.load(args.weight, map_location="cpu")
        net.load_weight(ckpt.get('net', ckpt))
    else:
        synthetic.load_seeded(net, seed=cfg.config['seed'])
    net = net.cuda().eval()
    if args.demo_path:
        data = io.load_demo_data(args.demo_path, cfg.config['data']['num_point'], seed=cfg.config['seed'])
    else:
        pc = synthetic.synthetic_scene(seed=args.synthetic if args.synthetic is not None else 10,
                                       n_points=cfg.config['data']['num_point'])
        data = {'point_clouds': torch.from_numpy(pc[None])}
    data = {k: v.cuda() for k, v in data.items()}
    torch.cuda.synchronize()
    t0 = time.time()
    end_points, ids, meshes = net.generate(data, selection=args.selection)
    torch.cuda.synchronize()
    print('Time elapsed: %s.' % (time.time() - t0))                       # demo.py:411
    box = keep = None
    if 'parsed_predictions' in end_points:
        box = end_points['parsed_predictions']['box_params'][0].cpu().numpy()
        keep = np.zeros(box.shape[0], dtype=bool)
        keep[ids[0, :, 0].cpu().numpy()] = True
    io.save_visualization(args.out, data['point_clouds'].cpu().numpy(), ids[0].cpu().numpy(), meshes, box, keep)
    print('%d meshes written to %s' % (len(meshes), args.out))


if __name__ == "__main__":
    main()
