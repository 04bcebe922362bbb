# -*- coding: utf-8 -*-
"""east -- host-side mirror of the EAST surface for the MI355X backend.

Same module and symbol names as the reference package for the hot path
(east.asts.base.AST.get_ast / score, east.relevance.ASTRelevanceMeasure,
east.applications.keyphrases_table, the `east keyphrases table` CLI); the
work runs in hand-written HIP kernels behind include/east_hip.h.
"""
from east import utils

utils.import_modules_from_package("east.asts")     # reference east/__init__.py:1-3
