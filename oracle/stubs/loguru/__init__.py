"""Test-only stand-in for `loguru` (absent from this image): a logger whose methods do nothing.
Used only to import the unmodified reference modules when generating / checking golden vectors."""


class _Logger:
    def __getattr__(self, name):
        def _noop(*args, **kwargs):
            return None

        return _noop


logger = _Logger()
