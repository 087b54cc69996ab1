#!/usr/bin/env python3
"""Evaluation loop over FewshotRegReader items, in the shape of the reference's driver
(test_rpnet.py:151-258: volumes -> 2-slice batches -> net(...) -> Dice per refinement
iteration), using only the symbols that driver imports, at the reference's import paths.
The reference file itself cannot run offline (it needs tensorboard and the private data).

    python tools/eval_driver.py --yaml yamls/example.yml [--items 2]
"""
import argparse
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dataset.few_shot_reader import FewshotRegReader
from net.model import model_factory
from net.registration import NCC
from utils.util import dice_score_seperate, load_yaml


def evaluate(net, loader, config, n_items=None, batch_size=2):
    net.eval()
    classes = config["eval_classes"]
    dsc_affine, dsc_fewshot, dsc_ref = defaultdict(list), defaultdict(list), defaultdict(lambda: defaultdict(list))
    for j in range(len(loader) if n_items is None else min(n_items, len(loader))):
        s = loader[j]
        with torch.no_grad():
            si = [[x.float().cuda() for x in way] for way in s["support_images"]]
            fg = [[x.float().cuda() for x in way] for way in s["support_labels"]]
            bg = [[1 - x for x in way] for way in fg]
            qi, ql, appr = s["query_images"].float().cuda(), s["query_labels"].long().cuda(), s["appr_query_labels"].cuda()
            fewshot, ref = [], defaultdict(list)
            for i in range(int(np.ceil(len(qi) / batch_size))):
                sl = slice(i * batch_size, (i + 1) * batch_size)
                out = net([[x[sl] for x in way] for way in si], [[x[sl] for x in way] for way in fg],
                          [[x[sl] for x in way] for way in bg], [qi[sl]], grid=s["grid"][sl], query_labels=ql[sl],
                          appr_query_labels=appr[sl])
                fewshot.append(out["output"].softmax(dim=1)[:, [1]].cpu())
                for k, v in out["refinement"].items():
                    ref[k].append(v.softmax(dim=1)[:, 1].cpu())
            pred = (torch.cat(fewshot, 0).permute(1, 0, 2, 3).numpy() > 0.5).astype(np.float32)
            gt = ql.cpu().numpy()[None]
            name = classes[s["class_id"]]
            d_aff = dice_score_seperate(appr.cpu().numpy()[None], gt, num_class=1)[0]
            d_few = dice_score_seperate(pred, gt, num_class=1)[0]
            ncc = NCC(qi, s["warped_supp"].unsqueeze(1).cuda()).item()
            dsc_affine[name].append(d_aff)
            dsc_fewshot[name].append(d_few)
            line = f"{j} {s['pid']} affine ({ncc:.4f}) {d_aff}, fewshot {d_few}"
            for k, v in ref.items():
                d = dice_score_seperate((torch.cat(v, 0).numpy() > 0.5).astype(np.int32)[None], gt, num_class=1)[0]
                dsc_ref[name][k].append(d)
                line += f" ref {k} {d},"
            print(line)
    for name in classes:
        if dsc_fewshot[name]:
            print(f"{name}, affine {np.mean(dsc_affine[name]):.4f}, fewshot {np.mean(dsc_fewshot[name]):.4f}")
    return dsc_affine, dsc_fewshot, dsc_ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--yaml", default="yamls/example.yml")
    ap.add_argument("--items", type=int, default=None)
    a = ap.parse_args()
    config, args = load_yaml(a.yaml)
    config["n_iter_refinement"] = config["n_test_iter_refinement"]            # test_rpnet.py:51
    loader = FewshotRegReader(args.data_dir, args.eval_set_name, config, mode="eval")
    net = model_factory[args.net](pretrained_path=config.get("pretrained_path"),
                                  cfg={"align": True, "backbone": config.get("backbone", "vgg")}, backbone_cfg=config).cuda()
    if args.ckpt:
        state = net.state_dict()
        state.update(torch.load(args.ckpt)["state_dict"])
        net.load_state_dict(state)
    evaluate(net, loader, config, a.items)


if __name__ == "__main__":
    main()
