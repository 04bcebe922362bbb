# -*- coding: utf-8 -*-
"""Exceptions of the EAST surface (reference east/exceptions.py:6-64)."""


class EastException(Exception):
    """Base EAST exception: `msg_fmt` is %-formatted with the constructor kwargs."""
    msg_fmt = "An unknown exception occurred."

    def __init__(self, message=None, **kwargs):
        self.kwargs = kwargs
        if not message:
            try:
                message = self.msg_fmt % kwargs
            except KeyError:
                message = self.msg_fmt
        super(EastException, self).__init__(message)

    def format_message(self):
        return str(self)


class NotFoundException(EastException):
    msg_fmt = "Not found."


class NoSuchASTAlgorithm(NotFoundException):
    msg_fmt = "There is no AST construction algorithm with name `%(name)s`."


class EmptyStringsCollectionException(EastException):
    msg_fmt = "The input strings collection is empty."


class SymbolOutOfDomainException(EastException):
    """New: the reference's string terminators are the code points U+0A00+i (east/asts/utils.py:25-40), so
    its own symbol encoding cannot hold text at or above U+0A00.  Raised where that encoding is asked for
    explicitly (asts/utils.py: strings_to_symbols(tagged=False)); the HIP backend itself indexes such text
    in the tagged encoding."""
    msg_fmt = ("Text%(where)s contains the code point U+%(code)04X, which collides with the "
               "string terminators of the annotated suffix tree (outside the method's domain).")

    def __init__(self, code, document=None):
        """document: index (or name) of the first text the character was found in, if known."""
        self.code, self.document = code, document
        super(SymbolOutOfDomainException, self).__init__(
            code=code, where="" if document is None else " %s" % (repr(document) if isinstance(document, str) else "number %d" % document))


class HipBackendError(EastException):
    """New: the MI355X backend is unavailable or failed.  There is no CPU fallback."""
    msg_fmt = "HIP backend error: %(reason)s"
