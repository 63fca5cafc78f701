"""/root/reference/utils/system_utils.py"""
import os


def mkdir_p(folder_path):
    os.makedirs(folder_path, exist_ok=True)


def searchForMaxIteration(folder):
    return max(int(name.split("_")[-1]) for name in os.listdir(folder))
