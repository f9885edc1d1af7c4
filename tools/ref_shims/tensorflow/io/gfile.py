import glob as _glob
import os
import shutil

exists = os.path.exists
GFile = open


def makedirs(p):
  os.makedirs(p, exist_ok=True)


def copy(a, b, overwrite=False):
  shutil.copyfile(a, b)


def rename(a, b, overwrite=False):
  os.replace(a, b)


remove = os.remove
glob = _glob.glob
