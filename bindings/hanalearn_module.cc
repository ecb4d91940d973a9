// The compiled `hanalearn` module on libhsad.so: same names and constructor signatures as the reference's cpp/pybind.cc:14-56.
#include "hsad_host.h"

namespace py = pybind11;
using namespace hsadpy;

PYBIND11_MODULE(hanalearn, m) {
  m.doc() = "hanalearn (cpp/pybind.cc) on the MI355X device pipeline: libhsad.so behind the reference's class names";
  py::module_::import("rela");      // ThreadLoop / R2D2Actor are registered there, like in the reference
  py::class_<HanabiEnv, std::shared_ptr<HanabiEnv>>(m, "HanabiEnv")
      .def(py::init<const std::unordered_map<std::string, std::string>&, const std::vector<float>&, int, bool, bool, bool, bool>())
      .def("feature_size", &HanabiEnv::feature_size)
      .def("num_action", &HanabiEnv::num_action)
      .def("hand_feature_size", &HanabiEnv::hand_feature_size)
      .def("terminated", &HanabiEnv::terminated)
      .def("get_current_player", &HanabiEnv::get_current_player)
      .def("last_score", &HanabiEnv::last_score)
      .def("get_score", &HanabiEnv::get_score)
      .def("get_life", &HanabiEnv::get_life)
      .def("get_info", &HanabiEnv::get_info)
      .def("get_fireworks", &HanabiEnv::get_fireworks);

  py::class_<HanabiVecEnv, std::shared_ptr<HanabiVecEnv>>(m, "HanabiVecEnv")
      .def(py::init<>())
      .def("append", &HanabiVecEnv::append, py::keep_alive<1, 2>())
      .def("size", &HanabiVecEnv::size);

  py::class_<HanabiThreadLoop, ThreadLoop, std::shared_ptr<HanabiThreadLoop>>(m, "HanabiThreadLoop")
      .def(py::init([](std::shared_ptr<R2D2Actor> a, std::shared_ptr<HanabiVecEnv> v, bool eval) {
        return std::make_shared<HanabiThreadLoop>(std::vector<std::shared_ptr<R2D2Actor>>{std::move(a)}, std::move(v), eval, false);
      }))
      .def(py::init([](std::vector<std::shared_ptr<R2D2Actor>> a, std::shared_ptr<HanabiVecEnv> v, bool eval) {
        return std::make_shared<HanabiThreadLoop>(std::move(a), std::move(v), eval, true);
      }))
      .def("num_games", [](HanabiThreadLoop& l) { return l.env ? l.env->G : 0; })
      .def("merged", [](HanabiThreadLoop& l) { return l.absorbed; })
      .def("check_errors", &HanabiThreadLoop::check_errors);
}
