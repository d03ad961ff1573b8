from .accuracy_calculator import CustomCalculator, get_accuracy_calculator
from .get_knn import get_knn
from . import hamming

__all__ = ["CustomCalculator", "get_accuracy_calculator", "get_knn", "hamming"]
