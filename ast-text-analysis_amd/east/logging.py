# -*- coding: utf-8 -*-
"""One-line progress display on an interactive stdout (same entry points as the reference's
east/logging.py: progress(message, step, total) and clear())."""
import sys

from east import utils

_LINE_WIDTH = 80


def _interactive():
    return not utils.output_is_redirected()


def _rewrite_line(text):
    sys.stdout.write("\r" + text)
    sys.stdout.flush()


def progress(message, step, total):
    """Overwrite the current line with `message: step/total` unless stdout is redirected."""
    if _interactive():
        _rewrite_line("%s: %i/%i" % (message, step, total))


def clear():
    """Blank the progress line."""
    if _interactive():
        _rewrite_line(" " * _LINE_WIDTH + "\r")


def warning(message):
    """A line on stderr (new here: the reference has no warnings of its own)."""
    sys.stderr.write("east: warning: %s\n" % message)
