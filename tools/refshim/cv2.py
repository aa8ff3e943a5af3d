"""Placeholder so that the reference's modules, which `import cv2` at the top, can be imported where OpenCV is not installed
(test tooling only).  Every attribute access fails loudly: nothing the fixture generator runs may depend on cv2."""


def __getattr__(name):
    raise AttributeError(f"refshim cv2 placeholder: cv2.{name} was used, but OpenCV is not installed here")
