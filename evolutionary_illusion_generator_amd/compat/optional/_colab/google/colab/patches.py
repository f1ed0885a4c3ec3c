"""Stand-in for `from google.colab.patches import cv2_imshow` (generate_illusion.py:7) outside Colab: the reference
shows best_flow.png / enhanced.png inline after every generation (:659, :673); off Colab there is nothing to show."""


def cv2_imshow(image):
    shape = getattr(image, "shape", None)
    print("cv2_imshow: image %s (not displayed outside Colab)" % (shape,))
