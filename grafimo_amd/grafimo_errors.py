"""Exception types raised through exception_handler (mirrors grafimo_errors.py:9-86 by name)."""


class GrafimoError(Exception):
    pass


class NotValidMotifMatrixError(GrafimoError):
    pass


class MotifProcessingError(GrafimoError):
    pass


class BGFileError(GrafimoError):
    pass


class FileReadError(GrafimoError):
    pass


class MotifFileReadError(GrafimoError):
    pass


class MotifFileFormatError(GrafimoError):
    pass


class NotValidFFormatError(GrafimoError):
    pass


class SubprocessError(GrafimoError):
    pass


class VGError(GrafimoError):
    pass


class FileFormatError(GrafimoError):
    pass
