"""Data-set / config cases shared by tests/golden/gen_golden_reader.py (reference side) and tests/test_volume_reader.py."""

CASES = {
    # crop larger than the truncated slice (symmetric padding), k blocks < depth
    "pad": {"seed": 4100, "data": dict(n_volumes=3, classes=("Liver",), shape=(22, 44, 40), seed=11),
            "cfg": dict(num_slice=20, num_x=36, num_y=40, crop_size=[48, 48], k=4), "registration": [0, 2]},
    # crop smaller than the slice, k larger than the annotated depth (clamps), odd sizes
    "cut": {"seed": 4200, "data": dict(n_volumes=4, classes=("Liver", "Spleen"), shape=(19, 50, 46), seed=12),
            "cfg": dict(num_slice=280, num_x=272, num_y=272, crop_size=[32, 32], k=12), "registration": [1]},
}


def config_for(case, class_csv_dir):
    cfg = dict(class_csv_dir=class_csv_dir, train_classes=list(case["data"]["classes"]), eval_classes=list(case["data"]["classes"]),
               n_shot=1, n_way=1, pad_value=-1024, HU_range=[-1024, 3072], do_elastic=True, do_intaug=True,
               gamma_range=[0.5, 1.5], use_registration_loss=True, use_registration_mask=True, do_deformable=False)
    cfg.update(case["cfg"])
    return cfg
