"""The three `utils.util` symbols test_rpnet.py imports (test_rpnet.py:15,28,30), written
fresh without the reference module's pydicom / SimpleITK / skimage / cv2 import chain
(/root/reference/utils/util.py:4-28 — none of it is used on this path)."""
import sys

import yaml


class Logger(object):
    """tee for sys.stdout (reference utils/util.py:63-76): everything printed also goes to `logfile`."""

    def __init__(self, logfile):
        self.terminal = sys.stdout
        self.log = open(logfile, "a")

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)

    def flush(self):
        self.terminal.flush()
        self.log.flush()


class _Struct:
    def __init__(self, **entries):
        self.__dict__.update(entries)


def load_yaml(path):
    """yaml -> (dict, attribute view of the same keys) (reference utils/util.py:79-88);
    duplicate keys: the last one wins, as with yaml.FullLoader in the reference."""
    with open(path) as f:
        data = yaml.load(f, Loader=yaml.FullLoader)
    return data, _Struct(**data)


def dice_score_seperate(y_pred, y_true, num_class=1, decimal=4):
    """Per-class Dice 2|P∩T| / (|P|+|T|), None for a class absent from the ground truth
    (reference utils/util.py:379-390).  y_pred / y_true: [num_class, ...] binary arrays."""
    scores = []
    for c in range(num_class):
        t, p = y_true[c], y_pred[c]
        if t.sum():
            scores.append(round(float(2 * (t * p).sum() / float(t.sum() + p.sum())), decimal))
        else:
            scores.append(None)
    return scores


def pad2factor(image, factor=16, pad_value=0):
    """[D,H,W] padded at the far end of every axis up to the next multiple of `factor` (reference utils/util.py:406-419)."""
    import numpy as np
    widths = [(0, -s % factor) for s in image.shape]
    return np.pad(image, widths, "constant", constant_values=pad_value)


def normalize(img, minimum=-1024, maximum=3076):
    """HU window -> [-1, 1] (reference utils/util.py:455-467): clip at the 99.5th percentile of THIS array, then at
    [minimum, maximum], scale by max(1, maximum - minimum).  Returns a new array of the input dtype family."""
    import numpy as np
    out = np.array(img, copy=True)
    top = float(np.percentile(out, 100.0 - 0.5))
    out[out > top] = top
    out[out > maximum] = maximum
    out[out < minimum] = minimum
    out = (out - minimum) / max(1, (maximum - minimum))
    return out * 2 - 1
