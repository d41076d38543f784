"""The package's logger.  The reference hands ONE logger to its helper (`RAGHelperLocal(logger)`, server/server.py:134-146) and every
component logs through it (`self.logger.info(...)`, server/RAGHelper.py:381-500); `factory.from_env(..., logger=logger)` (or
`set_logger`) makes the hot-path objects do the same.  Without one, records go to `logging.getLogger("ragmeup_amd")` -- warnings and
errors still reach stderr through logging's last-resort handler."""
from __future__ import annotations

import logging

_logger: logging.Logger | None = None


def set_logger(logger) -> None:
    """Route the package's records through `logger` (anything with .info / .warning / .error); None restores the default."""
    global _logger
    _logger = logger


def get_logger():
    return _logger if _logger is not None else logging.getLogger("ragmeup_amd")
