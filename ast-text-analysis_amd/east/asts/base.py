# -*- coding: utf-8 -*-
"""The plugin seam of the annotated-suffix-tree backends.

Contract (the one the reference defines in east/asts/base.py:10-46, which callers such as
east/relevance.py:44-53 rely on):

* a backend is a concrete subclass of `AST` carrying the class attribute `__algorithm__`;
  importing its module registers it (east/__init__.py imports every module of east.asts)
* `AST.get_ast(strings_collection, ast_algorithm="easa")` builds the backend registered under
  that name; unknown name -> NoSuchASTAlgorithm, empty collection -> EmptyStringsCollectionException
* every backend answers `score(query, normalized, synonimizer, return_suffix_scores)` and the
  three traversals that `traverse(callback, order)` dispatches to
"""
import abc
import inspect

from east import consts
from east import exceptions
from east import utils


class AST(abc.ABC):
    """Abstract annotated suffix tree over a collection of strings."""

    # ---- factory ---------------------------------------------------------------
    @classmethod
    def backends(cls):
        """{algorithm name: backend class} of everything registered so far."""
        table = {}
        for candidate in utils.itersubclasses(AST):
            name = getattr(candidate, "__algorithm__", None)
            if name is not None and not inspect.isabstract(candidate) and name not in table:
                table[name] = candidate
        return table

    @classmethod
    def get_ast(cls, strings_collection, ast_algorithm="easa"):
        try:
            backend = cls.backends()[ast_algorithm]
        except KeyError:
            raise exceptions.NoSuchASTAlgorithm(name=ast_algorithm)
        return backend(strings_collection)

    def __init__(self, strings_collection):
        if len(strings_collection) == 0:
            raise exceptions.EmptyStringsCollectionException()

    # ---- what a backend implements ------------------------------------------------
    @abc.abstractmethod
    def score(self, query, normalized=True, synonimizer=None, return_suffix_scores=False):
        """Matching score of `query`; with return_suffix_scores also {suffix: its score}."""

    @abc.abstractmethod
    def traverse_depth_first_pre_order(self, callback):
        """Visit every node, a parent before its children."""

    @abc.abstractmethod
    def traverse_depth_first_post_order(self, callback):
        """Visit every node, the children before their parent."""

    @abc.abstractmethod
    def traverse_breadth_first(self, callback):
        """Visit every node level by level."""

    # ---- dispatch -----------------------------------------------------------------
    def traverse(self, callback, order=consts.TraversalOrder.DEPTH_FIRST_PRE_ORDER):
        """Run the traversal named by `order` (one of consts.TraversalOrder); other values do nothing."""
        routes = ((consts.TraversalOrder.DEPTH_FIRST_PRE_ORDER, self.traverse_depth_first_pre_order),
                  (consts.TraversalOrder.DEPTH_FIRST_POST_ORDER, self.traverse_depth_first_post_order),
                  (consts.TraversalOrder.BREADTH_FIRST, self.traverse_breadth_first))
        for name, visit in routes:
            if name == order:
                visit(callback)
                return
