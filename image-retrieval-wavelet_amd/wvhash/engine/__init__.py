from .accuracy_calculator import CustomCalculator, get_accuracy_calculator
from .evaluate import evaluate, evaluate_multi_k, get_tester
from .get_knn import get_knn
from . import hamming

__all__ = ["CustomCalculator", "get_accuracy_calculator", "evaluate", "evaluate_multi_k", "get_tester",
           "get_knn", "hamming"]
