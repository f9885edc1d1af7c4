from logging import *  # noqa
import logging as _l

exception = _l.exception
