"""Constants and the error funnel the hot path depends on (utils.py:19-32, :63-78)."""
import sys

import numpy as np

DNA_ALPHABET = ["A", "C", "G", "T"]
REV_COMPL = {"A": "T", "C": "G", "G": "C", "T": "A"}
UNIF = "unfrm_dst"  # sentinel of -k/--bgfile meaning "uniform background" (utils.py:23)
PSEUDOBG = np.double(0.0000005)
LOG_FACTOR = 1.44269504
RANGE = 1000
SOURCE = "grafimo"
TP = "nucleotide_motif"
PHASE = "."


def die(code):
    sys.exit(code)


def exception_handler(exception_type, exception, debug):
    """Reference error behaviour (utils.py:63-78): with --debug raise
    ``exception_type("\\n\\n" + msg)``; otherwise print ``ERROR: msg`` to stderr
    and exit with status 1."""
    if debug:
        raise exception_type("\n\n{}".format(exception))
    sys.stderr.write("\n\nERROR: " + "{}".format(exception) + "\n")
    die(1)


def isListEqual(lst1, lst2):
    return len(lst1) == len(lst2) and set(lst1) == set(lst2)


def almost_equal(value1, value2, slope):
    return not ((value1 - slope) > value2 or (value1 + slope) < value2)


def print_progress_bar(iteration, total, prefix="", suffix="", decimals=1, length=50,
                       fill="=", print_end="\r"):
    """Same text as the reference's bar (utils.py:607-652)."""
    percent = ("{0:." + str(decimals) + "f}").format(100 * (iteration / float(total)))
    filled = int(length * iteration // total)
    bar = fill * filled + " " * (length - filled)
    print("\r%s [%s] %s%% %s" % (prefix, bar, percent, suffix), end=print_end)
    if iteration == total:
        print()
